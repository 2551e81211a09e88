"""Host-side pieces of cama_amd.engine that need no GPU."""
import builtins
import sys

import numpy as np

from cama_amd import engine


def test_content_hasher_falls_back_to_hashlib_without_xxhash(monkeypatch):
    """Engine.shared_map keys device maps on their content; xxhash is optional (requirements.txt) -- without it the key
    comes from hashlib.blake2b, still 128 bits and still a function of the bytes only."""
    data = np.arange(1000, dtype=np.float32).view(np.uint8).data
    real_import = builtins.__import__

    def no_xxhash(name, *a, **k):
        if name == "xxhash":
            raise ImportError("blocked for the test")
        return real_import(name, *a, **k)

    monkeypatch.delitem(sys.modules, "xxhash", raising=False)
    monkeypatch.setattr(builtins, "__import__", no_xxhash)
    h1, h2 = engine._content_hasher(), engine._content_hasher()
    assert type(h1).__module__.startswith("_blake2") or "blake2" in type(h1).__name__.lower()
    h1.update(data)
    h2.update(data)
    assert h1.hexdigest() == h2.hexdigest() and len(h1.hexdigest()) == 32
    h3 = engine._content_hasher()
    h3.update(np.arange(1, 1001, dtype=np.float32).view(np.uint8).data)
    assert h3.hexdigest() != h1.hexdigest()


def test_requirements_file_lists_what_the_default_path_imports():
    import os
    req = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "requirements.txt")).read()
    for name in ("numpy", "scipy", "PyYAML", "tqdm", "torch", "Pillow"):
        assert name in req


def test_chunked_mosaic_indexes_frames_and_refuses_slices_across_chunks():
    """engine.ChunkedMosaic: a long clip's mosaic as one allocation per launch (Engine.alloc_mosaics).  Frames by int,
    views by slices inside one chunk; ClipManager.render_clip cuts its launches at `bounds`."""
    import pytest
    import torch
    chunks = [torch.full((n, 2, 4, 3), k, dtype=torch.uint8) for k, n in enumerate((4, 4, 3))]
    m = engine.ChunkedMosaic(chunks)
    assert m.shape == (11, 2, 4, 3) and len(m) == 11 and m.bounds == [0, 4, 8, 11]
    assert [int(m[f][0, 0, 0]) for f in range(11)] == [0] * 4 + [1] * 4 + [2] * 3 and int(m[-1][0, 0, 0]) == 2
    v = m[4:8]
    assert v.shape[0] == 4 and v.data_ptr() == chunks[1].data_ptr()
    assert m[9:11].data_ptr() == chunks[2][1:].data_ptr() and m[:3].shape[0] == 3 and m[5:5].shape[0] == 0
    with pytest.raises(IndexError):
        m[3:5]
    with pytest.raises(IndexError):
        m[11]
    with pytest.raises(IndexError):
        m[0:8:2]
    m.fill_(7)
    assert all(int(c.min()) == 7 and int(c.max()) == 7 for c in chunks)
    assert [(lo, hi) for lo, hi, _ in m.spans()] == [(0, 4), (4, 8), (8, 11)]


def test_placement_helpers_only_probe_what_the_probe_kernel_can_read():
    """Engine.alloc_mosaic / place_frames / alloc_mosaics fall back to plain allocations unless the source is a contiguous
    uint8 [F, C, H, W, 3] tensor of the rig's size with W % 16 == 0 and opaque stamps (what cama_overlay_probe takes)."""
    import types
    import torch
    eng = types.SimpleNamespace(alpha256=256)
    rig = types.SimpleNamespace(C=6, H=4, W=32)
    ok = torch.zeros((3, 6, 4, 32, 3), dtype=torch.uint8)
    probe = engine.Engine._probeable
    assert probe(eng, rig, ok)
    assert probe(eng, rig, ok[1:])                                   # a slice along frames stays contiguous
    assert not probe(eng, rig, ok[:, :, :, ::2])                     # strided view
    assert not probe(eng, rig, ok.permute(0, 1, 3, 2, 4))            # wrong layout
    assert not probe(eng, rig, ok.to(torch.int8))
    assert not probe(eng, types.SimpleNamespace(C=6, H=4, W=24), torch.zeros((3, 6, 4, 24, 3), dtype=torch.uint8))   # W % 16
    assert not probe(types.SimpleNamespace(alpha256=128), rig, ok)   # translucent stamps go through another kernel
    assert not probe(eng, rig, torch.zeros((3, 6, 4, 32), dtype=torch.uint8))
