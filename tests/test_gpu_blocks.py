"""GPU unit tests of the hand-written building blocks, through the C ABI (bit-exact: integer work)."""
import ctypes as C
import importlib

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
L = importlib.import_module("3dgs_hierarchical_training_amd._lib")


@pytest.fixture(params=[0, 2], ids=["three-kernel", "onesweep"], autouse=True)
def sort_algo(request):
    lib = L.load()
    lib.gsr_set_option(b"sort_algo", request.param)
    yield request.param
    lib.gsr_set_option(b"sort_algo", 2)


def _sort(keys, vals, bits, u16=False):
    lib = L.load()
    dev = torch.device("cuda:0")
    n = keys.numel()
    k, v = keys.to(dev).clone(), vals.to(dev).clone()
    ka, va = torch.empty_like(k), torch.empty_like(v)
    sb = lib.gsr_sort_scratch_bytes(n)
    scratch = torch.empty(sb, dtype=torch.uint8, device=dev)
    in_alt = C.c_int(0)
    fn = lib.gsr_sort_pairs_u16 if u16 else lib.gsr_sort_pairs_u32
    L.check(fn(k.data_ptr(), v.data_ptr(), ka.data_ptr(), va.data_ptr(), n, 0, bits, scratch.data_ptr(), sb,
               C.byref(in_alt), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "sort")
    torch.cuda.synchronize()
    return (ka, va) if in_alt.value else (k, v)


@pytest.mark.parametrize("n", [1, 63, 4096, 4097, 100003, 1 << 20])
@pytest.mark.parametrize("bits,few", [(32, False), (32, True), (16, False), (8, False)])
def test_radix_sort_u32_stable(n, bits, few):
    g = torch.Generator().manual_seed(n + bits)
    hi = 7 if few else (1 << bits)
    keys = torch.randint(0, hi, (n,), generator=g, dtype=torch.int64)
    vals = torch.arange(n, dtype=torch.int64)
    sk, sv = _sort(keys.to(torch.int32) if bits < 32 else (keys - (keys >= 2**31) * 2**32).to(torch.int32), vals.to(torch.int32), bits)
    order = torch.sort(keys, stable=True).indices
    assert torch.equal(sv.cpu().to(torch.int64), order), "values not in stable sorted order"
    got_keys = sk.cpu().to(torch.int64) & 0xFFFFFFFF
    assert torch.equal(got_keys, keys[order])


@pytest.mark.parametrize("n,bits", [(1_500_000, 32), (5_000_000, 32), (2_000_003, 17), (700_001, 20), (3_000_000, 24)])
def test_radix_sort_u32_large_and_odd_widths(n, bits):
    """More workgroups than the chip holds at once (ticketed order; 1.5 M keys sit between one and two residencies of the
    1024-thread pass kernel) and the digit splits of wide tile keys: 17 bits = 3 x 6 (64-entry tables), 20 bits = 3 x 7,
    24 bits = 3 x 8."""
    g = torch.Generator().manual_seed(n + bits)
    keys = torch.randint(0, 1 << bits, (n,), generator=g, dtype=torch.int64)
    vals = torch.arange(n, dtype=torch.int64)
    sk, sv = _sort((keys - (keys >= 2**31) * 2**32).to(torch.int32), vals.to(torch.int32), bits)
    order = torch.sort(keys, stable=True).indices
    assert torch.equal(sv.cpu().to(torch.int64), order), "values not in stable sorted order"
    assert torch.equal(sk.cpu().to(torch.int64) & 0xFFFFFFFF, keys[order])


@pytest.mark.parametrize("n", [5, 4096, 300001, 6_000_000])
@pytest.mark.parametrize("T", [200, 2170, 40000])
def test_radix_sort_u16_stable(n, T):
    g = torch.Generator().manual_seed(n + T)
    keys = torch.randint(0, T, (n,), generator=g, dtype=torch.int64)
    vals = torch.arange(n, dtype=torch.int64)
    bits = 8 if T <= 256 else 16
    sk, sv = _sort(keys.to(torch.int16) if T < 32768 else (keys - (keys >= 32768) * 65536).to(torch.int16), vals.to(torch.int32), bits, u16=True)
    order = torch.sort(keys, stable=True).indices
    assert torch.equal(sv.cpu().to(torch.int64), order)
    assert torch.equal(sk.cpu().to(torch.int64) & 0xFFFF, keys[order])


def test_exports_and_sizes():
    lib = L.load()
    for s in L.EXPORTS:
        assert hasattr(lib, s)
    assert lib.gsr_geom_bytes(1000) >= 48 * 1000
    assert lib.gsr_image_bytes(980, 545) >= 980 * 545 * 28

