"""Runs tests/host/test_adapters on the GPU box: the AcquisitionInterface adapters of gnss-sdr_amd/host/gnss_sdr_adapters/ (GPS L1 C/A,
Galileo E1, GPS L5; gr_complex and cshort items; make_two_steps) built against the reference's own interface headers and driven
through the GNU Radio general_work contract until they publish their "events" message.  The binary is built by
__graft_entry__.build() in the container that has /root/reference (it cannot be built on the GPU box) and travels with the tree."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "host", "test_adapters")


@pytest.mark.gpu
def test_gnss_sdr_adapters_end_to_end(gpu):
    if not os.path.exists(BIN):
        import __graft_entry__ as g
        if not g.build_adapter_test():
            pytest.skip("tests/host/test_adapters was not prebuilt and /root/reference is not present here")
    r = subprocess.run([BIN], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ADAPTERS OK" in r.stdout, r.stdout[-4000:] + r.stderr[-2000:]
    _check_acquisition_dump(r.stdout)
    _check_acquisition_dump_two_steps(r.stdout)


def _check_acquisition_dump_two_steps(stdout):
    """make_two_steps (acq.cc:392-400): acq_grid is the wide grid of step one, acq_grid_narrow the second_nbins columns of step two, doppler_step_narrow and
    doppler_grid_narrow_min place its columns -- and the block's reported Doppler is the narrow grid's arg-max column, truncated to an integer (acq.cc:436)."""
    import numpy as np
    import scipy.io as sio
    line = [ln for ln in stdout.splitlines() if ln.startswith("ACQ_DUMP2 ")][-1].split()
    path, delay, doppler = line[1], float(line[3]), float(line[5])
    m = sio.loadmat(path)
    assert {"acq_grid", "acq_grid_narrow", "doppler_step_narrow", "doppler_grid_narrow_min"} <= set(m)
    g, gn = m["acq_grid"], m["acq_grid_narrow"]
    assert g.shape == (4000, 40) and gn.dtype == np.float32 and gn.shape == (4000, 8), (g.shape, gn.shape)
    assert float(m["doppler_step_narrow"][0, 0]) == 62.5
    tau1, bin1 = np.unravel_index(int(np.argmax(g)), g.shape)
    centre = -5000 + 250 * bin1                                   # step one's Doppler: the centre of the narrow grid (acq.cc:619)
    assert float(m["doppler_grid_narrow_min"][0, 0]) == pytest.approx(centre - 4 * 62.5)
    tau2, bin2 = np.unravel_index(int(np.argmax(gn)), gn.shape)
    assert gn.max() > 0.0 and float(tau2) == pytest.approx(delay) and tau1 == tau2
    assert float(int(centre + (bin2 - 4) * 62.5)) == pytest.approx(doppler)
    # the wide grid is step one's, whole: its column of the centre bin is not the narrow grid's first column
    assert not np.array_equal(g[:, 0], gn[:, 0])


def _check_acquisition_dump(stdout):
    """pcps_acquisition::dump_results (acq.cc:354-406): the .mat file the dumped channel left behind (MAT-file level 5, host/hip_mat5_writer.h) holds the
    reference's variables with its classes and dimensions, and what it holds is the search the block reported."""
    import numpy as np
    import scipy.io as sio
    line = [ln for ln in stdout.splitlines() if ln.startswith("ACQ_DUMP ")][-1].split()
    path, delay, doppler, stamp = line[1], float(line[3]), float(line[5]), int(line[7])
    m = sio.loadmat(path)
    names = {"acq_grid", "doppler_max", "doppler_step", "positive_acq", "acq_doppler_hz", "acq_delay_samples", "test_statistic", "threshold", "input_power",
             "sample_counter", "PRN", "num_dwells"}
    assert {k for k in m if not k.startswith("__")} == names
    g = m["acq_grid"]
    assert g.dtype == np.float32 and g.shape == (4000, 2 * 5000 // 250), g.shape   # effective_fft_size x Doppler bins (acq.cc:118: ceil(2 doppler_max / doppler_step)), one column per bin
    tau, dbin = np.unravel_index(int(np.argmax(g)), g.shape)
    assert m["positive_acq"][0, 0] == 1 and m["PRN"][0, 0] == 14 and m["num_dwells"][0, 0] == 1
    assert m["doppler_max"].dtype == np.int32 and m["doppler_max"][0, 0] == 5000 and m["doppler_step"][0, 0] == 250
    assert float(m["acq_delay_samples"][0, 0]) == pytest.approx(delay) and float(tau) == pytest.approx(delay)
    assert float(m["acq_doppler_hz"][0, 0]) == pytest.approx(doppler) and -5000 + 250 * dbin == pytest.approx(doppler)
    assert m["sample_counter"].dtype == np.uint64 and int(m["sample_counter"][0, 0]) == stamp
    assert float(m["test_statistic"][0, 0]) > float(m["threshold"][0, 0]) > 0.0
    # the CFAR statistic is the grid's maximum over the input power (acq.cc:438-445)
    assert float(g.max()) / float(m["input_power"][0, 0]) == pytest.approx(float(m["test_statistic"][0, 0]), rel=1e-5)
