"""fp32 fast mode (RR_DTYPE_F32) on the GPU against the REFERENCE's fp64 vectors and against the fp64 oracle
(BASELINE config 2: "4,096 parallel arenas ... fp32 state"; north_star: "fp32 positions within 1e-5").

Scoring is in tests/fp32_checks.py: single step from synchronised state; QUIET steps (pure kinematics) must meet
|delta| <= 1e-5 * max(1, |x|) on every position (centre) and velocity, 1e-5 of the 360-degree range on the rotation, 1e-5 *
max(10, |x|) on the derived AABB edges (why: fp32_checks.score), with integer state and done exact; steps with contact responses are held to the documented error distribution:
median at fp32 round-off, a thin tail where a response amplifies it, < 2.5 % branch flips."""
import numpy as np
import pytest

import fp32_checks as fc
import oracle_lib as ol

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

BAR = 1e-5


def _env(preset, n, **kw):
    import roborugby_amd as rr
    kw.setdefault("time_limit", False)
    kw.setdefault("auto_reset", False)
    return rr.BatchedRoboRugbyEnv(n, preset=preset, dtype="f32", **kw)


def _assert_distribution(tag, q, e, ints, done_ok):
    c = ~q
    assert q.sum() > 100 and c.sum() > 100, (tag, int(q.sum()), int(c.sum()))
    assert done_ok.all(), tag
    # quiet steps: the north-star bar, integer state exact
    assert e[q].max() <= BAR, (tag, float(e[q].max()))
    assert ints[q].all(), tag
    # contact steps: the documented distribution
    med, p90 = float(np.median(e[c])), float(np.percentile(e[c], 90))
    flips = float((e[c] > 1e-2).mean())
    assert med < 1e-5 and p90 < 3e-4 and flips < 0.025 and ints[c].mean() > 0.995, (tag, med, p90, flips, float(ints[c].mean()))
    print(f"[{tag}] fp32 vs fp64: {int(q.sum())} quiet steps max rel err {e[q].max():.2e} (bar {BAR:g}); {int(c.sum())} contact steps "
          f"median {med:.2e} p90 {p90:.2e} p99 {np.percentile(e[c], 99):.2e}, {100 * flips:.2f} % > 1e-2, integer state equal in "
          f"{100 * ints[c].mean():.2f} %")


@pytest.mark.parametrize("preset", ["T", "G"])
def test_f32_single_step_vs_reference_golden(golden_dir, preset):
    """Every recorded reference step (tests/golden/traj_*.npz) replayed through an RR_DTYPE_F32 env from the reference's
    dumped state."""
    t = np.load(f"{golden_dir}/traj_{preset}.npz")
    cfg = ol.PRESETS[preset]
    idx = [(ep, s) for ep in range(t["length"].shape[0]) for s in range(int(t["length"][ep]))]
    ep = np.array([i[0] for i in idx]); s = np.array([i[1] for i in idx])
    pre = {k: t["state_" + k][ep, s] for k in ("robots", "robots_i", "balls", "step")}
    post = {k: t["state_" + k][ep, s + 1] for k in ("robots", "robots_i", "balls", "step")}
    acts = t["actions"][ep, s]
    na_used = (acts >= 0).sum(1)
    n = len(idx)
    got = {k: np.zeros_like(post[k]) for k in ("robots", "robots_i", "balls")}
    done_ok = np.zeros(n, bool)
    for k in np.unique(na_used):
        sel = np.nonzero(na_used == k)[0]
        env = _env(preset, len(sel))
        env.set_state(pre["robots"][sel], pre["robots_i"][sel], pre["balls"][sel], pre["step"][sel])
        o, r, d, info = env.step(torch.as_tensor(acts[sel][:, :k].astype(np.int32)))
        st = env.get_state()
        for kk in got:
            got[kk][sel] = st[kk].cpu().numpy()
        done_ok[sel] = d.cpu().numpy().astype(np.uint8) == t["done"][ep[sel], s[sel]]
        assert np.array_equal(st["step"].cpu().numpy(), post["step"][sel])
        env.close()
    q, e, ints = fc.score(pre, post, got, cfg["W"], cfg["H"])
    _assert_distribution(f"{preset} golden", q, e, ints, done_ok)


@pytest.mark.parametrize("preset", ["T", "G"])
def test_f32_config2_rollout_vs_f64_oracle(preset):
    """BASELINE config 2: 4,096 arenas, fp32 state, 25-step random-policy rollout.  The fp64 oracle free-runs each arena
    from the kernel's own reset; before every step the fp32 env is synchronised to the oracle's state, so each of the
    4,096 x 25 steps is a single-step comparison (chaotic dynamics: free-running fp32 and fp64 drift apart by design)."""
    n, steps, seed = 4096, 25, 5
    cfg = ol.PRESETS[preset]
    env = _env(preset, n, seed=seed)
    env.reset()
    st = {k: v.cpu().numpy() for k, v in env.get_state().items()}
    na = env.preset.nr
    orcs = []
    for a in range(n):
        o = ol.OracleEnv(preset)
        o.reset(seed, a, 0); o.reset(seed, a, 1)  # constructor placement, then env.reset(): the kernel's own stream
        orcs.append(o)
    ost = [o.get_state() for o in orcs]
    # same integer draws: centres and rotations identical, edges to fp32 round-off
    assert np.array_equal(np.array([s_["robots"][:, [0, 1, 6]] for s_ in ost]), st["robots"][:, :, [0, 1, 6]])
    rng = np.random.default_rng(99)
    Q, E, I, D = [], [], [], []
    for k in range(steps):
        pre = {kk: np.array([s_[kk] for s_ in ost]) for kk in ("robots", "robots_i", "balls")}
        pre_step = np.array([s_["step"] for s_ in ost], np.int32)
        env.set_state(pre["robots"], pre["robots_i"], pre["balls"], pre_step)
        acts = rng.integers(0, 8, size=(n, na)).astype(np.int32)
        o32, r32, d32, info = env.step(torch.as_tensor(acts))
        res = [orcs[a].step(acts[a]) for a in range(n)]
        ost = [o.get_state() for o in orcs]
        post = {kk: np.array([s_[kk] for s_ in ost]) for kk in ("robots", "robots_i", "balls")}
        got = {kk: v.cpu().numpy() for kk, v in env.get_state().items()}
        ok = np.array([(r_["status"] & 63) == 0 for r_ in res])  # steps in which the reference would have raised are not scored
        q, e, ints = fc.score(pre, post, got, cfg["W"], cfg["H"])
        Q.append(q[ok]); E.append(e[ok]); I.append(ints[ok])
        D.append((d32.cpu().numpy() == np.array([r_["done"] for r_ in res]))[ok])
        # observations / rewards are float32 outputs: scored on the quiet arenas only (lidar rays nearly parallel to a side are ill-conditioned)
        oo = np.array([r_["obs"] for r_ in res])
        oe = fc.rel_err(o32.cpu().numpy().astype(np.float64)[ok & q], oo[ok & q])
        assert np.percentile(oe, 99) < 2e-3, float(np.percentile(oe, 99))
    q, e, ints, dn = np.concatenate(Q), np.concatenate(E), np.concatenate(I), np.concatenate(D)
    assert dn.all()
    assert e[q].max() <= BAR and ints[q].all(), (preset, float(e[q].max()))
    c = ~q
    # random actions rarely touch anything: the non-quiet steps here are mostly robots near walls / each other, not responses
    assert np.median(e[c]) < 1e-5 and (e[c] > 1e-2).mean() < 0.025
    print(f"[{preset} config 2] {len(e)} fp32 steps vs fp64 oracle: quiet {int(q.sum())} max rel err {e[q].max():.2e} (bar {BAR:g}); "
          f"other {int(c.sum())}: median {np.median(e[c]):.2e} p99 {np.percentile(e[c], 99):.2e}, {100 * (e[c] > 1e-2).mean():.2f} % > 1e-2")
