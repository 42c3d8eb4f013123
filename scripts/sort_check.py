#!/usr/bin/env python3
"""Standalone check of the queue-ordering step against numpy's stable argsort."""
import importlib, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
rt = importlib.import_module("rust-raytracer_amd")
ctx = rt.Context(0)
rng = np.random.default_rng(1)
bad = False
for n in (1, 63, 4096, 4353, 49152, 300000, 2_000_000, 16_588_800):
    for bits in (21, 32):
        keys = rng.integers(0, 1 << 30, size=n, dtype=np.uint32)
        keys[rng.random(n) < 0.1] = 0xFFFFFFFF
        perm = ctx.debug_sort_keys(keys, bits)
        keys = keys & np.uint32((0xFFFFFFFF << (32 - bits)) & 0xFFFFFFFF)   # callers zero the bits they do not sort by
        ref = np.argsort(keys >> np.uint32(32 - bits), kind="stable").astype(np.uint32)
        ok = np.array_equal(perm, ref)
        is_perm = np.array_equal(np.sort(perm), np.arange(n, dtype=np.uint32))
        mk = (keys >> np.uint32(32 - bits))
        sorted_masked = is_perm and bool(np.all(np.diff(mk[perm].astype(np.int64)) >= 0))
        sorted_full = is_perm and bool(np.all(np.diff(keys[perm].astype(np.int64)) >= 0))
        print("n=%d bits=%d %s  permutation=%s sorted(masked)=%s sorted(full key)=%s max=%d" % (
            n, bits, "ok" if ok else "MISMATCH", is_perm, sorted_masked, sorted_full, int(perm.max())), flush=True)
        bad = bad or not (is_perm and sorted_masked)
print("sort check passed" if not bad else "sort check FAILED")
sys.exit(1 if bad else 0)
