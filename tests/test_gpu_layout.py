"""GPU: the force-directed layout steps of Graph::postprocess (rala_hip_layout, reference
graph.cpp:1132-1226) - bit-identical to the sequential CPU evaluation (FP64, same order of
operations), alone and inside the clean-up pipeline against the oracle restatement."""
import numpy as np
import pytest

import layout
import test_layout_cpu as cpu

pytestmark = pytest.mark.gpu


def _hip_engine(ctx):
    def engine(x, y, adj_off, adj, iterations, k, t, dt):
        ctx.layout(x, y, adj_off, adj, iterations, k, t, dt)
        return 0
    return engine


@pytest.mark.parametrize("n,seed", [(1, 0), (7, 1), (255, 2), (256, 3), (257, 4), (1000, 5), (3001, 6)])
def test_layout_steps_match_sequential_evaluation(hip_ctx_factory, n, seed):
    rng = np.random.default_rng(seed)
    x, y = rng.random(n), rng.random(n)
    if n > 3:
        x[1], y[1] = x[0], y[0]                        # coincident points: the 0.01 clamp
        x[2], y[2] = x[0] + 1e-4, y[0]
    deg = rng.integers(0, 6, size=n)
    adj_off = np.zeros(n + 1, dtype=np.uint32)
    np.cumsum(deg, out=adj_off[1:])
    adj = rng.integers(0, n + 1, size=int(adj_off[n])).astype(np.uint32)       # n = the origin
    k = float(np.sqrt(1.0 / n))
    t, dt, iterations = 0.1, 0.1 / 101, 50 if n <= 1000 else 6
    wx, wy = x.copy(), y.copy()
    layout.numpy_engine(wx, wy, adj_off, adj, iterations, k, t, dt)
    gx, gy = x.copy(), y.copy()
    hip_ctx_factory().layout(gx, gy, adj_off, adj, iterations, k, t, dt)
    assert (gx == wx).all() and (gy == wy).all(), (np.abs(gx - wx).max(), np.abs(gy - wy).max())


@pytest.mark.parametrize("seed", range(8))
def test_layout_in_the_pipeline(hip_ctx_factory, seed):
    ctx = hip_ctx_factory()
    graphs = cpu._both()
    cpu._tangle(graphs, seed)
    graphs[0].postprocess(seed, _hip_engine(ctx))
    graphs[1].postprocess(seed)
    layout.assert_same_graph(*graphs, "after postprocess")
    assert (graphs[0].edge_weights() > 0).any()
    la, lb = [], []
    cpu._simplify(graphs[0], la, _hip_engine(ctx))
    cpu._simplify(graphs[1], lb)
    assert la == lb
    layout.assert_same_graph(*graphs, "after simplify")
