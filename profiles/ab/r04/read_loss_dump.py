import sys, glob, numpy as np
LOG = np.dtype([("VE", "<f4"), ("E", "<f4"), ("P", "<f4"), ("L", "<f4"), ("VL", "<f4"), ("prompt_I", "<f4"), ("prompt_Q", "<f4"), ("start", "<u8"), ("acc_phase", "<f4"), ("doppler", "<f4"),
                ("doppler_rate", "<f4"), ("code_freq", "<f4"), ("code_freq_rate", "<f4"), ("carr_err", "<f4"), ("carr_err_filt", "<f4"), ("code_err", "<f4"), ("code_err_filt", "<f4"),
                ("cn0", "<f4"), ("lock_test", "<f4"), ("aux1", "<f4"), ("aux2", "<f8"), ("PRN", "<u4"), ("TOW_ms", "<u8"), ("WN", "<i4")])
for f in glob.glob(sys.argv[1] + "*"):
    r = np.fromfile(f, dtype=LOG)
    print(f, len(r), "records")
    bad = [i for i in range(len(r)) if not np.isfinite([r[i][k] for k in ("P", "prompt_I", "doppler", "code_freq", "cn0", "lock_test")]).all()]
    print("first non-finite record:", bad[:3])
    for i in list(range(200, 212)) + list(range(1010, 1024)):
        if i < len(r):
            print(i, int(r["start"][i]), "d_start", int(r["start"][i]) - int(r["start"][i - 1]), "P %.1f I %.1f Q %.1f dop %.2f codef %.2f cn0 %.2f lock %.3f" % (r["P"][i], r["prompt_I"][i], r["prompt_Q"][i], r["doppler"][i], r["code_freq"][i], r["cn0"][i], r["lock_test"][i]))
