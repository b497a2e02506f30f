// Device memory through HIP's virtual-memory API (hipMemCreate / hipMemAddressReserve / hipMemMap) for tools/aux_vmm.py: one
// physical allocation of exactly the requested size mapped at a fresh address, instead of whatever hipMalloc's pools hand out.
//   hipcc -shared -fPIC tools/probes/vmm_alloc.cpp -o tools/probes/libvmm.so
#include <hip/hip_runtime.h>
#include <cstdio>

extern "C" {

// bytes is rounded up to the allocation granularity (returned through *granularity); NULL on failure (reason on stderr)
void *vmm_alloc(size_t bytes, size_t *granularity, size_t *mapped_bytes)
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return nullptr;
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = dev;
    size_t gran = 0;
    hipError_t e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended);
    if (e != hipSuccess || gran == 0) { std::fprintf(stderr, "vmm: granularity: %s\n", hipGetErrorString(e)); return nullptr; }
    const size_t size = (bytes + gran - 1) / gran * gran;
    hipMemGenericAllocationHandle_t h;
    if ((e = hipMemCreate(&h, size, &prop, 0)) != hipSuccess) { std::fprintf(stderr, "vmm: hipMemCreate: %s\n", hipGetErrorString(e)); return nullptr; }
    void *ptr = nullptr;
    if ((e = hipMemAddressReserve(&ptr, size, gran, nullptr, 0)) != hipSuccess) { std::fprintf(stderr, "vmm: reserve: %s\n", hipGetErrorString(e)); return nullptr; }
    if ((e = hipMemMap(ptr, size, 0, h, 0)) != hipSuccess) { std::fprintf(stderr, "vmm: map: %s\n", hipGetErrorString(e)); return nullptr; }
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    if ((e = hipMemSetAccess(ptr, size, &acc, 1)) != hipSuccess) { std::fprintf(stderr, "vmm: access: %s\n", hipGetErrorString(e)); return nullptr; }
    (void)hipMemRelease(h);                       // the mapping keeps the physical memory alive
    if (granularity) *granularity = gran;
    if (mapped_bytes) *mapped_bytes = size;
    return ptr;
}

int vmm_free(void *ptr, size_t mapped_bytes)
{
    hipError_t e = hipMemUnmap(ptr, mapped_bytes);
    if (e == hipSuccess) e = hipMemAddressFree(ptr, mapped_bytes);
    return (int)e;
}

}  // extern "C"
