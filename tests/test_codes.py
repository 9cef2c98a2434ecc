"""gnss_sdr_amd/codes.py (the bench's own C/A code source) against the reference's generator as compiled into the oracle."""
import numpy as np
import pytest

import oracle


def test_ca_codes_equal_the_reference_generator():
    from gnss_sdr_amd.codes import gps_l1_ca_code, gps_l1_ca_code_sampled
    for prn in range(1, 33):
        assert np.array_equal(gps_l1_ca_code(prn), oracle.ca_code(prn)), prn
    for fs in (4000000, 25000000, 2048000, 16368000, 8000000):
        for prn in (1, 7, 32):
            assert np.array_equal(gps_l1_ca_code_sampled(prn, fs), oracle.ca_code_complex_sampled(prn, fs)), (prn, fs)
    with pytest.raises(ValueError):
        gps_l1_ca_code(33)
