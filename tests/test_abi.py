"""CPU tests of the C-ABI boundary: the shared library loads, exports every symbol the header declares, and
refuses to compute without a GPU (there is no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from raytracer_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    import glob
    headers = sorted(glob.glob(os.path.join(ROOT, "include", "*.h")))
    assert [os.path.basename(h) for h in headers] == ["rt_camera.h", "rt_dist.h", "rt_hotpath.h", "rt_overlay.h"]
    text = "".join(open(h).read() for h in headers)
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    L = api.lib()
    names = declared_symbols()
    assert len(names) >= 14
    for n in names:
        assert hasattr(L, n), n
    assert sorted(api.EXPORTED_SYMBOLS) == names


def test_struct_layouts_match_header():
    assert C.sizeof(api.rt_uniforms) == 280
    assert C.sizeof(api.rt_frame) == 280 + 8 * 4 + 3 * 8
    assert C.sizeof(api.rt_target) == 8 + 6 * 8
    assert C.sizeof(api.rt_stats) == 80


def test_argument_validation_without_compute():
    L = api.lib()
    h = C.c_void_p()
    assert L.rt_scene_create(None, None, 0, None, 0, None, 0, None, None, C.byref(h)) == api.RT_EINVAL
    assert b"bounds" in L.rt_last_error()
    assert L.rt_render(None, None, None) == api.RT_EINVAL
    assert L.rt_scene_destroy(None) == api.RT_OK
    z = np.zeros(3, np.float32)
    bad_ref = np.array([[0, 1, 9, 0]], np.uint32)   # vertex index out of range
    v = np.zeros((3, 4), np.float32)
    assert L.rt_scene_create(v.ctypes.data, v.ctypes.data, 3, bad_ref.ctypes.data, 1, None, 0, z.ctypes.data, z.ctypes.data, C.byref(h)) == api.RT_ESCENE
    bad_mat = np.array([[0, 1, 2, 7]], np.uint32)   # material id beyond the shader's table
    assert L.rt_scene_create(v.ctypes.data, v.ctypes.data, 3, bad_mat.ctypes.data, 1, None, 0, z.ctypes.data, z.ctypes.data, C.byref(h)) == api.RT_ESCENE


@pytest.mark.skipif(api.device_count() > 0, reason="a GPU is present")
def test_no_cpu_fallback():
    z = np.zeros(3, np.float32)
    with pytest.raises(api.RtError) as e:
        api.Scene(np.zeros((0, 4)), np.zeros((0, 4)), np.zeros((0, 4)), np.zeros(0), z, z)
    assert e.value.code == api.RT_ENODEVICE
    with pytest.raises(api.RtError) as e:      # the device builder behind it has no host stand-in either
        api.Scene.from_meshes([], device=0)
    assert e.value.code == api.RT_ENODEVICE


def test_one_hip_runtime_in_the_process():
    """The library and PyTorch must share one copy of the HIP runtime whatever is imported first (a second copy finds no
    GPU): checked in a fresh interpreter that loads the library before it imports torch."""
    import subprocess
    import sys
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from raytracer_amd import api\n"
            "api.lib()\n"
            "import torch\n"
            "libs = sorted({l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l})\n"
            "print(len(libs), libs)\n" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    assert out.stdout.split()[0] == "1", out.stdout
