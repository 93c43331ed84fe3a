"""CPU-side checks of the drop-in boundary: the shared library loads, exports every symbol
include/sots_hip.h declares, validates configurations, and refuses to compute without a
gfx950 device (there is no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "sots_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sots_[a-z_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(hip):
    lib = hip.load()
    names = declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"libsots_hip.so does not export {n}"
    assert sorted(hip.EXPORTS) == names


def test_config_struct_layout_matches_header(hip):
    # 10 uint32/int32 + uint64 + 2 x float[16]
    assert C.sizeof(hip.Config) == 10 * 4 + 8 + 2 * 16 * 4
    assert hip.Config.seed.offset == 40 and hip.Config.param_min.offset == 48


def _cfg(hip, **over):
    cfg = hip.Config()
    cfg.struct_size = C.sizeof(hip.Config)
    cfg.num_parents, cfg.num_offspring, cfg.num_dimensions = 16, 16, 4
    cfg.audio_length_log2, cfg.synth_kind, cfg.workgroup_size = 10, 0, 32
    cfg.device, cfg.seed = 0, 1
    for k, v in over.items():
        setattr(cfg, k, v)
    return cfg


@pytest.mark.parametrize("over,needle", [
    (dict(struct_size=12), "struct_size"),
    (dict(synth_kind=9), "synth_kind"),
    (dict(num_dimensions=6), "numDimensions"),
    (dict(audio_length_log2=7), "audioLengthLog2"),
    (dict(audio_length_log2=16), "audioLengthLog2"),
    (dict(num_parents=0), "population"),
    (dict(workgroup_size=0), "workgroupSize"),
    (dict(workgroup_size=24), "workgroupSize"),
])
def test_create_rejects_bad_configs(hip, over, needle):
    lib = hip.load()
    h = C.c_void_p()
    rc = lib.sots_create(C.byref(_cfg(hip, **over)), C.byref(h))
    assert rc == -1 and not h.value
    assert needle in lib.sots_last_error(None).decode()


def test_null_arguments(hip):
    lib = hip.load()
    assert lib.sots_create(None, None) == -1
    assert lib.sots_execute_generation(None) == -1
    assert lib.sots_stage_fft(None) == -1
    lib.sots_destroy(None)  # harmless


def test_no_cpu_fallback_without_a_gpu(hip):
    """On a box without a GPU, a valid configuration must fail with SOTS_ERR_NO_DEVICE rather
    than silently computing somewhere else."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present; the GPU suite covers the success path")
    lib = hip.load()
    h = C.c_void_p()
    rc = lib.sots_create(C.byref(_cfg(hip)), C.byref(h))
    assert rc == -3 and not h.value
    with pytest.raises(hip.SotsError):
        hip.HipES(16, 16, 0, 10, None, [3520.0, 8.0, 3520.0, 1.0])


def test_group_create_validates_without_a_gpu(hip):
    """sots_group_create checks its arguments before it touches a device; without a gfx950 device a valid request
    fails with SOTS_ERR_NO_DEVICE like sots_create (no CPU fallback for islands either)."""
    lib = hip.load()
    g = C.c_void_p()
    one = (C.c_int32 * 1)(0)
    cfg = _cfg(hip)
    assert lib.sots_group_create(None, one, 1, 4, 1, 0, C.byref(g)) == -1
    assert lib.sots_group_create(C.byref(cfg), one, 0, 4, 1, 0, C.byref(g)) == -1 and b"numDevices" in lib.sots_group_last_error(None)
    many = (C.c_int32 * 17)(*range(17))
    assert lib.sots_group_create(C.byref(cfg), many, 17, 4, 1, 0, C.byref(g)) == -1
    assert lib.sots_group_create(C.byref(_cfg(hip, struct_size=8)), one, 1, 4, 1, 0, C.byref(g)) == -1
    rc = lib.sots_group_create(C.byref(cfg), one, 1, 4, 1, 0, C.byref(g))
    if rc == 0:      # a GPU is present (the GPU box): a one-island group is fine
        assert lib.sots_group_size(g) == 1 and not lib.sots_group_uses_rccl(g)
        lib.sots_group_destroy(g)
    else:
        assert rc == -3 and not g.value
    assert lib.sots_group_size(None) == 0 and lib.sots_group_island(None, 0) is None
