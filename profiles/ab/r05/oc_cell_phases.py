"""Stage clocks of the acquisition cell kernel oc_cell_kernel<Plan<25,25,40>,1,false,false,false> (BASELINE config 3: 32 PRN x 41 bins, N = 25 000).
Run ON THE GPU BOX with the library built with -DGSH_OC_PROFILE:
    python profiles/ab/build_variant.py ocprof pcps_onchip -DGSH_OC_PROFILE        (in the container; the .so travels)
    GSH_LIB_PATH=build/variants/lib_ocprof.so python profiles/ab/r05/oc_cell_phases.py > profiles/oc_cell_annotated.txt
Every wave of every cell leaves the shader clock at eight points and the 100 MHz wall clock at entry and exit (csrc/pcps_onchip.hip, OC_STAMP)."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import torch

import gnss_sdr_amd
from gnss_sdr_amd import _lib
from gnss_sdr_amd.acquisition import PcpsAcquisitionBank

n, fs = 25000, 25000000
dev = torch.device("cuda", 0)
x = torch.view_as_complex(torch.randn(n, 2, device=dev).contiguous())
acq = PcpsAcquisitionBank(fs_in=fs, fft_size=n, doppler_max=5000, doppler_step=250, num_doppler_bins=41, samples_per_chip=25, samples_per_code=float(n), max_prn=32, device=0,
                          keep_grid=False)
code = (np.random.randn(n) + 1j * np.random.randn(n)).astype(np.complex64)
for p in range(32):
    acq.set_local_code(p, code)
for _ in range(3):
    acq.time_dwells(x, 32, reps=20)            # clocks up
ms_single = min(acq.time_dwells(x, 32, reps=40) for _ in range(3))
ms_pipe = min(acq.time_dwells(x, 32, reps=100, pipelined=True) for _ in range(3))
acq.time_dwells(x, 32, reps=1)                 # the batch whose stamps are read
L = _lib.load()
W, WAVES, CELLS = 10, 16, 32 * 41
buf = (C.c_uint64 * (4096 * WAVES * W))()
L.gsh_debug_oc_profile.argtypes = [C.c_void_p, C.c_size_t]
L.gsh_debug_oc_profile.restype = C.c_int
assert L.gsh_debug_oc_profile(buf, len(buf)) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(4096, WAVES, W)[:CELLS].astype(np.int64)
clk, wall = a[:, :, :8], a[:, :, 8:]
# shader clocks per 100 MHz tick, from the cells' own entry / exit stamps
dur_clk = (clk[:, 0, 7] - clk[:, 0, 0]).astype(np.float64)
dur_wall = (wall[:, 0, 1] - wall[:, 0, 0]).astype(np.float64)
ghz = float(np.median(dur_clk / np.maximum(dur_wall, 1.0))) * 0.1   # clocks per 10 ns -> GHz
names = ["operands: 2 x 25 loads of 8 B per thread from L2 / Infinity Cache, conj(X) C",
         "stage 1: radix-25 butterflies in registers + twiddles",
         "exchange 1: LDS, whole complex values in 3 phases of 9 + 9 + 7 rows (write phase p, barrier, read phase p - 1 ...)",
         "stage 2: radix-25 butterflies + twiddles",
         "exchange 2: 3 phases again, no barrier in front (starts in the region exchange 1 did not end in)",
         "stage 3 (radix 40, 625 of the 1000 threads) + |.|^2 + per-thread max / arg-max / sum + wave reduction",
         "the end of the cell: " + os.environ.get("OC_END_NOTE", "the wave's partial record to memory; rows and statistic by oc_rows_kernel behind the cells")]
print("oc_cell_kernel<Plan<25,25,40>, 1, false, false, false> -- one work-group of 1000 threads (16 waves) per (PRN, Doppler bin) cell, 1312 cells per batch; "
      f"cells per work-group: {os.environ.get('GSH_OC_CELLS_PER_WG', '1')}")
print(f"this run: {ms_single * 1e3:.1f} us per batch alone, {ms_pipe * 1e3:.1f} us per batch with two batches in flight; shader clock {ghz:.2f} GHz (cells' own stamps)")
print()
print("clocks per stage, from each wave's stamps (s_memtime at the end of the stage; median over the 1312 cells):")
print(f"{'stage':<118} {'wave 0':>9} {'slowest wave':>13} {'fastest wave':>13}   us (wave 0)")
tot0 = 0.0
for k in range(7):
    d = (clk[:, :, k + 1] - clk[:, :, k]).astype(np.float64)       # cells x waves
    w0, slow, fast = np.median(d[:, 0]), np.median(d.max(axis=1)), np.median(d.min(axis=1))
    tot0 += w0
    print(f"{names[k]:<118} {w0:9.0f} {slow:13.0f} {fast:13.0f}   {w0 / ghz / 1e3:6.2f}")
life = (clk[:, :, 7] - clk[:, :, 0]).astype(np.float64)
print(f"{'a cell, entry to exit':<118} {np.median(life[:, 0]):9.0f} {np.median(life.max(axis=1)):13.0f} {np.median(life.min(axis=1)):13.0f}   {np.median(life[:, 0]) / ghz / 1e3:6.2f}")
# waves wait for each other at the barriers inside the exchanges: how far apart do they ARRIVE at each stamp?
print()
print("spread of the 16 waves' arrival at each stamp (latest - earliest, median over cells, clocks): what a barrier behind the stamp makes the early waves wait")
for k, nm in enumerate(["entry", "operands + products done", "stage 1 done", "exchange 1 done", "stage 2 done", "exchange 2 done", "stage 3 + wave reduction done", "exit"]):
    sp = (clk[:, :, k].max(axis=1) - clk[:, :, k].min(axis=1)).astype(np.float64)
    print(f"  {nm:<34} {np.median(sp):8.0f}")
# the batch's time line: cells per compute unit, gaps between a cell's exit and the next cell's entry on the same unit cannot be seen from here (no CU id),
# but the wall clock of the first entry and the last exit bound the kernel, and the sum of cell lives / 256 units is what a perfect packing would take
t0, t1 = wall[:, 0, 0].min(), wall[:, 0, 1].max()
span_us = (t1 - t0) * 0.01
cell_us = np.median(dur_wall) * 0.01
print()
print(f"time line: first cell entered, last cell left: {span_us:.1f} us; median cell {cell_us:.2f} us; 1312 cells / 256 compute units = 5.125 rounds -> "
      f"{5.125 * cell_us:.1f} us if every unit were busy all the time, {6 * cell_us:.1f} us for six whole rounds")
order = np.argsort(wall[:, 0, 0])
starts_us = (wall[order, 0, 0] - t0) * 0.01
print("cells entered by time (us since the first): " + ", ".join(f"{int((starts_us < t).sum())} by {t}" for t in (5, 20, 40, 60, 80, 100, 120)))
late = (wall[:, 0, 0] - t0) * 0.01 > 5 * cell_us
print(f"cells that start after {5 * cell_us:.0f} us (the sixth round): {int(late.sum())}")
# per wave: the waves 10 .. 15 have no stage-3 butterfly (radix 40: threads 640 .. 1 023) and load their own operands of the next cell there
print()
print("median clocks per stage and wave (rows: waves 0 .. 15; columns: operands, stage 1, exchange 1, stage 2, exchange 2, stage 3 / prefetch, end):")
for w in range(WAVES):
    d = [(clk[:, w, k + 1] - clk[:, w, k]).astype(np.float64) for k in range(7)]
    print(f"  wave {w:2d}  " + " ".join(f"{np.median(v):7.0f}" for v in d))
