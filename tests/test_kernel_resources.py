"""Register / scratch budget of the tracking kernels, read from the built library's code-object metadata (no GPU needed).

VERDICT round 1 asked for a closed-loop kernel without scratch (it had 976 B per thread); small changes elsewhere in the kernel have since pushed a few long-lived constants back
into scratch more than once (the 1 024-thread work-group leaves 128 VGPRs), so the property is pinned here.  The batched correlator's launch flavours must not spill either: a
spilling flavour ran 45 % slower (profiles/r02/mcorr_bound_experiments.txt)."""
import os
import re

import pytest

import gnss_sdr_amd
from kernel_metadata import kernels


@pytest.fixture(scope="module")
def meta():
    lib = gnss_sdr_amd._lib.LIB_PATH
    if not os.path.exists(lib):
        pytest.skip("library not built")
    k = kernels(lib)
    assert len(k) > 50, "no gfx950 code objects found in the library"
    return k


def test_closed_loop_kernels_use_no_scratch(meta):
    loop = {n: k for n, k in meta.items() if "trk_loop_kernel" in n}
    assert len(loop) >= 4
    for n, k in loop.items():
        assert k[".private_segment_fixed_size"] == 0, (n, k[".private_segment_fixed_size"])
        assert k[".vgpr_count"] <= 128, (n, k[".vgpr_count"])  # 1 024 threads per work-group


def test_batched_correlator_flavours_use_no_scratch(meta):
    # mcorr_kernel<NT, MODE, AUX, RUNS, WIN, PAIR>: every flavour except the run-based experiment (RUNS = true, behind GSH_MC_PACKED_BODY=2)
    pat = re.compile(r"mcorr_kernel(?:_t128)?ILi(\d)ELi(\d)ELb([01])ELb([01])ELb([01])ELb([01])E")
    seen = 0
    for n, k in meta.items():
        m = pat.search(n)
        if not m or m.group(4) == "1":
            continue
        seen += 1
        assert k[".private_segment_fixed_size"] == 0, (n, k[".private_segment_fixed_size"])
        if m.group(1) == "3" and m.group(2) == "0" and m.group(3) == "0":
            # E/P/L, standard mode: 6 waves per SIMD for the per-tap flavour (<= 80 VGPRs), 5 for the paired-tap flavour (<= 96)
            assert k[".vgpr_count"] <= (96 if m.group(6) == "1" else 80), (n, k[".vgpr_count"])
    assert seen >= 12


def test_on_chip_acquisition_kernels_use_no_scratch(meta):
    """Every flavour of the whole-transform-on-one-CU acquisition kernels (csrc/pcps_onchip.hip: oc_forward_kernel, oc_forward_split_kernel,
    oc_cell_kernel for every plan, split factor, grid / second-peak / offset switch).  Round 2 shipped ten flavours with 12 - 148 bytes of
    scratch per thread (the sub-cell kernels sit at the 128-register limit of a 1 024-thread work-group); VERDICT round 2 asked for none."""
    oc = {n: k for n, k in meta.items() if any(w in n for w in ("oc_cell_kernel", "oc_forward_kernel", "oc_forward_split_kernel", "oc_subcell_dit_kernel",
                                                                "oc_combine_dit_kernel", "oc_second_peak_kernel"))}
    assert len(oc) >= 180, len(oc)
    assert sum("oc_subcell_dit_kernel" in n for n in oc) >= 9 and sum("oc_combine_dit_kernel" in n for n in oc) >= 9  # the decimation-in-time pair (round 3)
    spilling = {n: k[".private_segment_fixed_size"] for n, k in oc.items() if k[".private_segment_fixed_size"] != 0}
    assert not spilling, spilling
    for n, k in oc.items():
        assert k[".vgpr_count"] <= 128, (n, k[".vgpr_count"])


def test_no_kernel_of_the_library_uses_scratch(meta):
    """EVERY kernel of libgnss_sdr_hip.so: the four-step acquisition kernels (csrc/pcps_fft.hip: fwd_cols_kernel, rows_kernel<0|1>,
    inv_cols_kernel) carried 496 B per thread -- the generic-radix butterfly's per-thread array of sums and differences, indexed at run time; it now re-reads the
    pass's input from LDS instead.  The run-based correlator experiment (16 - 32 B) is only compiled with -DGSH_MC_RUNS_EXPERIMENT."""
    spilling = {n: k[".private_segment_fixed_size"] for n, k in meta.items() if k[".private_segment_fixed_size"] != 0 and "ELb0ELb1ELb0ELb0EE" not in n}
    assert not spilling, spilling
    dynamic = [n for n, k in meta.items() if k.get(".uses_dynamic_stack")]
    assert not dynamic, dynamic
    for n, k in meta.items():
        wg = k.get(".max_flat_workgroup_size", 256)
        # a work-group of 1 024 threads leaves 128 registers per thread, one of 512 leaves 256
        assert k[".vgpr_count"] + k.get(".agpr_count", 0) <= 512 // max(1, (wg + 255) // 256), (n, k[".vgpr_count"], wg)
