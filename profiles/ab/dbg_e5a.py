import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from gnss_sdr_amd.detectors import GalileoE5aNoncoherentIQAcquisitionCaf
from oracle.pcps_oracle import E5aNoncoherentIqOracle
from detector_cases import e5a_case
x, kw, ci, cq = e5a_case(fs=8000000, sampled_ms=2, data_signs=(-1, 1), pilot_signs=(-1, 1), delay_chips=10.0, doppler=-1300.0)
n = kw["fft_size"]
o = E5aNoncoherentIqOracle(**kw); g = GalileoE5aNoncoherentIQAcquisitionCaf(device=0, **kw)
o.set_local_code(ci, cq); g.set_local_code(ci, cq)
o.work(x[:n]); g.work(x[:n])
print(o.result, g.result, float(o.mag), float(g.mag))
d = o.result["index_doppler"]
print("oracle row", o.rows[d]); print("gpu row", g.rows[d], g.slots)
for k, s in g.slots.items():
    pk, ix = g.bank.read_row_peaks(s)
    print(k, pk[d], ix[d])
d2 = g.result["index_doppler"]
print("oracle row at gpu bin", o.rows[d2]); print("gpu row", g.rows[d2])
