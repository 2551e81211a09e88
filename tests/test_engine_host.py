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
