import sys, numpy as np
sys.path.insert(0, '/root/repo')
from fft_amd import specengine as S
from oracle import spectrum_oracle as O
sys.path.insert(0, '/root/repo/tests')
from test_gpu_pipeline import make_params
worst_y = worst_m = 0
for n_base, up in [(2048, 8), (256, 8), (2048, 1)]:
    params = make_params(12, n_base + up, n_base)
    mag, pb, pm, y = S.synth_analyze(params, 2.0, n_base, up, "gaussian", return_signal=True)
    for i, p in enumerate(params):
        yu, mref, pbref, pmref = O.synth_upsample_analyze(p, 2.0, n_base, up, "gaussian")
        worst_y = max(worst_y, np.abs(y[i] - yu).max()); worst_m = max(worst_m, np.linalg.norm(mag[i] - mref) / np.linalg.norm(mref))
print("max |y err|", worst_y, "(bar 2e-5)  max rel-L2 mag", worst_m, "(bar 1.4e-4)")
