"""GPU parity tests proper: every C-ABI entry point against the oracle (see tests/gpu_checks.py)."""
import pytest

pytestmark = pytest.mark.gpu


def _assert_ok(res):
    bad = [r for r in res if not r["ok"]]
    assert not bad, "\n".join(f"{r['name']}: err={r['err']:.3e} scale={r['scale']:.3e} tol={r['tol']}" for r in bad)


@pytest.mark.parametrize("B,seed,K", [(3, 0, 4), (5, 2, 3), (1, 9, 4)])
def test_per_op_parity(B, seed, K):
    from tests import gpu_checks as G
    _assert_ok(G.run_all(B=B, seed=seed, K=K))


@pytest.mark.parametrize("B,seed,K,steps", [(3, 1, 4, 3), (7, 4, 2, 2)])
def test_fused_step_parity(B, seed, K, steps):
    from tests import gpu_checks as G
    _assert_ok(G.run_fused(B=B, seed=seed, K=K, steps=steps))
