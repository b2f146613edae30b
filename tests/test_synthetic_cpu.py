"""CPU: the product-side benchmark helpers (tpgsr_amd/utils/synthetic.py) follow the recipe the parity tests share with the reference
(oracle.recipe_state_dict / oracle.synthetic_batch), so a bench model is the same kind of network the parity suite checks -- and
bench.py builds its networks without importing the oracle."""
import os
import subprocess
import sys

import torch

from oracle import tpgsr_oracle as O

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_init_by_recipe_equals_the_oracle_recipe():
    from tpgsr_amd.model import tsrn
    from tpgsr_amd.model.crnn import crnn
    from tpgsr_amd.utils.synthetic import init_by_recipe
    for net, spec, seed, kw in ((tsrn.TSRN_TL(STN=True, mask=True), O.tsrn_spec(STN=True, mask=True, text_prior=True), 11, dict(tps_hw=(16, 64))),
                                (tsrn.TSRN(STN=True, mask=True), O.tsrn_spec(STN=True, mask=True), 1234, dict(tps_hw=(16, 64))),
                                (crnn.CRNN(32, 1, 37, 256), O.crnn_spec(), 12, {})):
        ref = O.recipe_state_dict(spec, seed, **kw)
        got = init_by_recipe(net, seed).state_dict()
        assert list(got.keys()) == list(ref.keys())
        for k in ref:
            assert got[k].shape == ref[k].shape and got[k].dtype == ref[k].dtype, k
            tol = 1e-5 if k.split(".")[-1] in ("inverse_kernel", "target_coordinate_repr", "padding_matrix", "target_control_points") else 0.0
            if k.endswith("stn_fc2.bias"):
                tol = 1e-7      # identity control points: float64 here, float32 in the oracle, then the same noise
            assert (got[k].double() - ref[k].double()).abs().max() <= tol, k


def test_synthetic_batch_equals_the_oracle_batch():
    from tpgsr_amd.utils.synthetic import synthetic_batch
    for n, seed in ((4, 1234), (3, 7)):
        a, b = synthetic_batch(n, seed), O.synthetic_batch(n, seed)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


def test_bench_builds_its_step_without_the_oracle():
    code = ("import sys; sys.path.insert(0, %r); import os; os.environ['TPGSR_PLAN_DRYRUN'] = '1'; import torch, bench; "
            "ts, nets = bench.build_step('c3', torch.device('cpu')); lr, hr = bench.synthetic_batch(2, 1, torch.device('cpu')); "
            "print('ORACLE_IMPORTED' if any(m == 'oracle' or m.startswith('oracle.') for m in sys.modules) else 'CLEAN')" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, timeout=300)
    assert out.stdout.strip().endswith("CLEAN"), out.stdout + out.stderr
