"""The reference's compiled-in configuration, read out of oracle/_ref/libdynaframe_static.so -- R/StaticParameters.cpp itself,
compiled where it lies by `make -C oracle ref` (the one translation unit of the reference that needs no OpenCV).  Test
infrastructure only; on a box without /root/reference the shared object travels as built, or is absent (the tests then fall
back on the committed fixture tests/golden/static_parameters.json, which was written from it)."""
import ctypes as C
import os

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_ref", "libdynaframe_static.so")
INTS = ("PROJECTOR_RESLINE", "PROJECTOR_RESROW", "CAMERA_RESLINE", "CAMERA_RESROW", "PC_BIASLINE", "PC_BIASROW", "GRAY_V_NUMDIGIT",
        "GRAY_H_NUMDIGIT", "PHASE_NUMDIGIT", "SHOW_PICTURE_TIME", "DYNAFRAME_MAXNUM", "FOV_MIN_DISTANCE", "FOV_MAX_DISTANCE", "RECO_WINDOW_SIZE")


def available():
    return os.path.exists(PATH)


def constants():
    """{name: int} of every `extern const int` of R/StaticParameters.h:8-42 (plus VISUAL_DEBUG as 0 / 1)."""
    lib = C.CDLL(PATH)
    out = {name: int(C.c_int.in_dll(lib, name).value) for name in INTS}
    out["VISUAL_DEBUG"] = int(C.c_bool.in_dll(lib, "VISUAL_DEBUG").value)
    return out
