#!/usr/bin/env python3
"""bench.py -- random-policy rollout throughput of the batched RoboRugby step on MI355X.

    python bench.py --gpus N --steps K --warmup W            (N>1: launched by torch.distributed.run)

A "step" is one pass of the hot path (GameEnv.step: 12 physics sub-steps + reward + observation + done,
reference RR_EnvBase.py:260-297) over one batch of 65,536 arenas per GPU.  Metric = BASELINE.json's
env-steps/s (whole job).  Inputs (state, actions) are resident in HBM when the timed region starts.
Rank 0 prints ONE JSON line, with `roofline` (dominant kernel vs the HBM roof) and `cpu_baseline`
(the CPU oracle timed on this box's host cores on a bounded sample -- a reported baseline, not the target).
"""
import argparse
import json
import os
import subprocess
import sys
import threading
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--arenas", type=int, default=65536, help="arenas per GPU (weak scaling)")
    ap.add_argument("--preset", default="G", choices=["G", "T", "D"],
                    help="G = constants as checked in (2+2 robots, 4+4 balls, 800x800); T = DQN training preset")
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32", "f32_state"])  # f32_state: fp32 records in HBM, fp64 arithmetic (BASELINE config 2)
    ap.add_argument("--policy", default="random", choices=["random", "chase"])
    ap.add_argument("--log-interval", type=int, default=25, help="steps between RCCL all-gathers of episode returns")
    ap.add_argument("--fuse", type=int, default=1, help="steps per launch (rr_rollout, open-loop extension; 1 = one launch per "
                    "step like the reference's gym API -- the headline)")
    ap.add_argument("--pipeline", type=int, default=1, help="extension, not the headline: the arenas as P independent shard envs "
                    "(arena_offset) stepped on P HIP streams, so that consecutive launches of different shards overlap and the chip "
                    "does not drain at the end of every step; random policy, one GPU")
    ap.add_argument("--budget", type=int, default=0, help="extension, not the headline: the budgeted step (rr_config.step_budget_clocks, "
                    "shader clocks): an arena over the budget after an expensive sub-step parks and reports NOT_READY; such rows are "
                    "NOT counted as env steps")
    ap.add_argument("--exact-trig", action="store_true", help="the exact-trig parity build (libroborugby_amd_exact.so): sin / cos of the "
                    "kinematics in double-double, ~correctly rounded; not the headline library")
    ap.add_argument("--no-stagger", action="store_true", help="time the steps right after a fresh reset (every arena at the same, "
                    "contact-poor episode phase).  Default: arenas at uniformly random episode phases, reached by a pre-roll of "
                    "one whole episode outside the timed region -- the steady state of a long rollout")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


def maybe_relaunch(args):
    """`python bench.py --gpus 4` without a launcher: start torch.distributed.run as a child (before any GPU use)."""
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", os.environ.get("MASTER_PORT", "29511"),
               os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))


def cpu_baseline(preset, seconds):
    """CPU oracle (oracle/rr_oracle.c: the fp64 restatement pinned bit-exact to the reference) stepping the same
    kind of workload -- random actions from reset -- on this box's host cores, one thread per core."""
    sys.path.insert(0, os.path.join(REPO, "tests"))
    import oracle_lib as ol
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))
    t0 = time.perf_counter()
    n_cal, _ = ol.rollout(preset, 2, 40, seed=99)
    rate1 = n_cal / max(time.perf_counter() - t0, 1e-6)
    steps = 100
    arenas = max(1, int(rate1 * seconds / steps))
    done = [0] * cores

    def work(i):
        done[i], _ = ol.rollout(preset, arenas, steps, seed=1000 + i)

    th = [threading.Thread(target=work, args=(i,)) for i in range(cores)]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    dt = time.perf_counter() - t0
    return {"value": sum(done) / dt, "unit": "env-steps/s", "cores": cores, "kind": "port",
            "sample": f"{cores} threads x {arenas} arenas x {steps} steps of preset {preset}, random actions from "
                      f"reset, fp64 C oracle (bit-exact to the Python reference on tests/golden); {dt:.1f} s wall",
            "single_core_value": rate1}


def hbm_copy_probe(dev, mib=1024, iters=20):
    """what a plain device-to-device copy reaches on THIS chip (read + write bytes / HIP-event time),
    quoted next to the 8 TB/s spec as roofline.peak_measured.  One GiB source, one GiB destination, far beyond the 256 MB of
    Infinity Cache; a few milliseconds in total, outside the timed region.  Two probes: the library's 16-B-per-lane grid-stride
    copy kernel (rr_probe_hbm_copy: the pattern MI355X_MICROARCH.md quotes ~6.3 TB/s for) and torch's copy_ (the runtime's D2D
    copy).  Returns (GB/s own kernel, GB/s torch copy_)."""
    import ctypes as C
    import torch
    from roborugby_amd import _lib
    lib = _lib.load()
    n = mib * 1024 * 1024 // 4
    src = torch.empty(n, dtype=torch.float32, device=dev).normal_()
    dst = torch.empty_like(src)
    st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    rates = []
    for mode in ("kernel", "torch"):
        def one():
            if mode == "kernel":
                _lib.check(lib.rr_probe_hbm_copy(C.c_void_p(dst.data_ptr()), C.c_void_p(src.data_ptr()), C.c_size_t(n * 4), st), "rr_probe_hbm_copy")
            else:
                dst.copy_(src)
        for _ in range(3):
            one()
        e0.record()
        for _ in range(iters):
            one()
        e1.record()
        torch.cuda.synchronize()
        rates.append(2.0 * n * 4 / (e0.elapsed_time(e1) / iters * 1e-3) / 1e9)
    assert torch.equal(dst, src)
    del src, dst
    return rates[0], rates[1]


def pipelined(args):
    """--pipeline P: the same 65,536 arenas, the same K steps each, as P shard envs on P streams (roborugby_amd.ShardedPipeline).
    A shard's step s+1 only waits for its own step s (and its own policy call), so while one shard's launch drains -- a launch
    ends with its slowest wavefront -- another's fills the chip.  Results per arena are the single batch's bit for bit
    (arena_offset keys the reset RNG; tests/test_gpu_shortcuts.py)."""
    import torch
    import roborugby_amd as rr
    assert args.gpus == 1 and args.fuse == 1 and args.arenas % args.pipeline == 0
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    P, N, K, W = args.pipeline, args.arenas, args.steps, args.warmup
    n = N // P
    pipe = rr.ShardedPipeline(N, shards=P, device=dev, preset=args.preset, seed=0, time_limit=True, auto_reset=True, dtype=args.dtype,
                              step_budget_clocks=args.budget)
    p = pipe.preset
    na = p.nr
    gens = [torch.Generator(device=dev) for _ in range(P)]
    for i, g in enumerate(gens):
        g.manual_seed(1234 + i)
    acts = torch.randint(0, 8, (K + W, N, na), generator=gens[0], device=dev, dtype=torch.int32) if args.policy == "random" else None
    outs = [(torch.empty(n, 11, device=dev), torch.empty(n, device=dev), torch.empty(n, dtype=torch.uint8, device=dev),
             torch.empty(n, 11, device=dev) if p.nr_grumpy else None, torch.empty(n, device=dev),
             torch.empty(n, dtype=torch.int32, device=dev)) for _ in range(P)]
    step_no = [0] * P

    def policy(i, obs):
        s = step_no[i]
        step_no[i] += 1
        if acts is not None:
            return acts[s, i * n:(i + 1) * n]
        from roborugby_amd import players
        return players.chase(pipe.envs[i], obs, step=s + 1, noise=0.1, seed=1234)

    obs0 = pipe.reset()
    for i in range(P):
        with torch.cuda.stream(pipe.streams[i]):
            outs[i][0].copy_(obs0[i])
    torch.cuda.synchronize()
    pipe.run(policy, W, outs=outs)
    torch.cuda.synchronize()
    cnt0 = sum(int(e.episode_stats()[3].sum().item()) for e in pipe.envs)
    # budgeted step: every call's status row is kept and the NOT_READY rows are subtracted afterwards (no extra launch)
    hist = [torch.zeros(K, n, dtype=torch.int32, device=dev) for _ in range(P)] if args.budget else None
    step_no[:] = [W] * P
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pipe.run(policy, K, outs=(lambda i, s: outs[i][:5] + (hist[i][s],)) if args.budget else outs)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    cnt1 = sum(int(e.episode_stats()[3].sum().item()) for e in pipe.envs)
    resets = cnt1 - cnt0 - sum(int(o[2].sum().item()) for o in outs)
    n_not_ready = sum(int(((h & 16384) != 0).sum().item()) for h in hist) if args.budget else 0
    steps = N * K - max(resets, 0) - n_not_ready
    bytes_per_step = p.algorithmic_bytes_per_step(na)
    achieved = bytes_per_step * steps / dt / 1e9
    lanes = pipe.envs[0].lanes_per_env()
    line = {"metric": "env_steps_per_sec", "value": steps / dt, "unit": "env-steps/s", "n_gpus": 1, "steps": K, "warmup": W,
            "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype,
            "data": "synthetic",
            "config": {"workload": f"{N} parallel arenas per GPU as {P} independent shards of {n} on {P} HIP streams (launches of "
                                   f"different shards overlap), SimpleDuel3 preset {args.preset}, {args.policy}-policy rollout, "
                                   f"{na} action(s)/arena, auto-reset on done, {lanes} lanes per arena",
                       "arenas_per_gpu": N, "preset": args.preset, "policy": args.policy, "lanes_per_arena": lanes,
                       "sharding": "single GPU", "steps_per_launch": 1, "pipeline": P, "step_budget_clocks": args.budget,
                       "not_ready_fraction": n_not_ready / float(N * K)},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": None, "kernel": "k_step", "kernel_ms": None,
                         "algorithmic_bytes_per_env_step": bytes_per_step, "record_bytes_per_env": pipe.envs[0].state_bytes_per_env(),
                         "note": "overlapping launches: achieved = algorithmic bytes of all arenas per step / wall time per step"}}
    print(json.dumps(line), flush=True)
    pipe.close()


def main():
    args = parse()
    if args.pipeline > 1:
        return pipelined(args)
    maybe_relaunch(args)
    import torch
    import roborugby_amd as rr
    from roborugby_amd import dist as rrd

    # device selection (incl. RR_SHARE_GPU: every rank on cuda:0, the one-GPU rehearsal of the N>1 path) happens inside,
    # before the process group is created, and RCCL's communicator is bound to that device
    rank, local_dev, world = rrd.init_process_group()
    dev = torch.device(f"cuda:{local_dev}")
    n = args.arenas
    env = rr.BatchedRoboRugbyEnv(n, preset=args.preset, device=dev, seed=0, time_limit=True, auto_reset=True,
                                 dtype=args.dtype, arena_offset=rrd.shard_offset(rank, n), step_budget_clocks=args.budget,
                                 exact_trig=args.exact_trig)
    p = env.preset
    na = p.nr
    obs = env.reset()
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    K, W = args.steps, args.warmup
    # action stream resident in HBM before the clock starts (the policy is not the thing measured)
    if args.policy == "random":
        acts = torch.randint(0, 8, (K + W, n, na), generator=gen, device=dev, dtype=torch.int32)
    out = (torch.empty(n, 11, device=dev), torch.empty(n, device=dev), torch.empty(n, dtype=torch.uint8, device=dev),
           torch.empty(n, 11, device=dev) if p.nr_grumpy else None, torch.empty(n, device=dev),
           torch.empty(n, dtype=torch.int32, device=dev))

    from roborugby_amd import players
    chase_out = torch.empty(n, na, dtype=torch.int32, device=dev)
    chase_step = [0]

    out[0].copy_(obs)

    def one_step(i, o=None):
        if args.policy == "random":
            a = acts[i]
        else:  # the scripted on-device policy (one launch, rr_policy_chase): robot 0 chases its ball, 10 % random, the rest random
            chase_step[0] += 1
            a = players.chase(env, out[0], step=chase_step[0], noise=0.1, seed=1234 + rank, out=chase_out)
        return env.step(a, out=out if o is None else o)

    stagger = not args.no_stagger and args.fuse == 1
    from_reset = None
    if stagger and args.policy == "random" and not args.budget and K + W < p.game_len_steps:
        # secondary number, same run: the K steps right after a fresh reset (what rounds 1-2 reported as the headline: every
        # arena at the same, contact-poor start of its episode) -- kept so that the lines stay comparable across rounds
        for i in range(W):
            env.step(acts[i], out=out)
        ev0 = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
        rrd.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(K):
            ev0[i][0].record()
            env.step(acts[W + i], out=out)
            ev0[i][1].record()
        torch.cuda.synchronize()
        rrd.barrier()
        dt0 = rrd.reduce_max(time.perf_counter() - t0, dev)
        k0 = rrd.reduce_max(sum(a.elapsed_time(b) for a, b in ev0) / K, dev)
        ach0 = p.algorithmic_bytes_per_step(na) * n / (k0 * 1e-3) / 1e9
        from_reset = {"value": rrd.reduce_sum(float(n * K), dev) / dt0, "ms_per_step": dt0 / K * 1e3, "kernel_ms": k0,
                      "roofline_achieved": ach0, "roofline_frac": ach0 / HBM_PEAK_GBS,
                      "note": f"{K} steps right after a fresh reset (steps {W + 1}..{W + K} of the episode), same process, before the "
                              "pre-roll: the definition of rounds 1-2 (r02: 184.2 M, frac 0.01415)"}
    if stagger:
        # Steady state: every arena at a uniformly random phase of its episode, with the state a rollout of that length leaves
        # (contacts accumulate late in an episode: robots park balls against walls).  The step counters are spread over
        # [0, T) and ONE whole episode is rolled outside the timed region, so each arena has been re-placed at its own time.
        st = env.get_state()
        T = p.game_len_steps
        phase = torch.randint(0, T, (n,), generator=gen, device=dev, dtype=torch.int32)
        env.set_state(st["robots"], st["robots_i"], st["balls"], phase)
        pre = torch.randint(0, 8, (min(T, 512), n, na), generator=gen, device=dev, dtype=torch.int32)
        for i in range(T + 1):
            if args.policy == "random":
                env.step(pre[i % pre.shape[0]], out=out)
            else:
                one_step(1)
        del pre
    F = max(1, args.fuse)
    if F > 1:
        assert args.policy == "random" and K % F == 0 and W % F == 0, "--fuse needs the random policy and K, W multiples of it"
        fout = (torch.empty(F, n, 11, device=dev), torch.empty(F, n, device=dev), torch.empty(F, n, dtype=torch.uint8, device=dev),
                torch.empty(F, n, 11, device=dev) if p.nr_grumpy else None, torch.empty(F, n, device=dev),
                torch.empty(F, n, dtype=torch.int32, device=dev))
    for i in range(0, W, F):
        if F > 1:
            env.rollout(acts[i:i + F], out=fout)
        else:
            one_step(i)
    _, _, _, cnt0 = env.episode_stats()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(K)]
    pending = []
    rrd.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    # the ONE collective of the path: finished-episode returns, all-gathered for logging every min(log_interval, K) steps
    # (the driver runs --steps 20: with the default interval of 25 the loop alone would never issue it)
    gather_every = max(F, min(args.log_interval, K) // F * F)
    status_hist = torch.zeros(K, n, dtype=torch.int32, device=dev) if args.budget else None
    for i in range(0, K, F):
        ev[i][0].record()
        if F > 1:
            env.rollout(acts[W + i:W + i + F], out=fout)  # F steps per launch
        elif args.policy == "random" and not args.budget:
            env.step(acts[W + i], out=out)  # k_step (+ the few-microsecond k_order that sorts the next dispatch) between the two events
        else:  # (budgeted: every call's status row is kept, the NOT_READY rows are counted after the loop -- no extra launch)
            one_step(W + i, out[:5] + (status_hist[i],) if args.budget else None)
        ev[i][1].record()
        if world > 1 and (i + F) % gather_every == 0:
            lr = env.episode_stats()[0]
            pending.append(rrd.all_gather_returns(lr, async_op=True))
    if world > 1:  # and once after the loop: the returns of the episodes that ended since the last interval
        pending.append(rrd.all_gather_returns(env.episode_stats()[0], async_op=True))
    for _, work in pending:
        if work is not None:
            work.wait()
    gathered_rows = [int(t.shape[0]) for t, _ in pending]
    torch.cuda.synchronize()
    rrd.barrier()
    dt = time.perf_counter() - t0
    dt = rrd.reduce_max(dt, dev)
    # steps that only re-placed a finished arena are not counted as env steps
    _, _, _, cnt1 = env.episode_stats()
    last_done = fout[2][-1] if F > 1 else out[2]  # arenas that finished in the very last step are re-placed by a later call
    resets = int((cnt1 - cnt0).sum().item()) - int(last_done.sum().item())
    n_not_ready = int(((status_hist & 16384) != 0).sum().item()) if args.budget else 0
    if args.budget:
        out = out[:5] + (status_hist[K - 1],)
    local_steps = n * K - max(resets, 0) - n_not_ready
    total_steps = rrd.reduce_sum(float(local_steps), dev)
    kern_ms = sum(ev[i][0].elapsed_time(ev[i][1]) for i in range(0, K, F)) / (K // F)  # per launch
    kern_ms = rrd.reduce_max(kern_ms, dev)
    status_bits = int(torch.bitwise_and((fout if F > 1 else out)[5], 0xFFFF & ~1024 & ~256 & ~16384).max().item())

    if rank == 0:
        bytes_per_step = p.algorithmic_bytes_per_step(na)            # G 601 B, T 149 B
        achieved = bytes_per_step * n * F / (kern_ms * 1e-3) / 1e9     # GB/s, one launch = n * F arena-steps
        traffic, traffic_source = None, None
        tfile = os.path.join(REPO, "profiles", "traffic.json")          # PMC-derived HBM bytes per launch, if measured
        if os.path.exists(tfile):
            try:
                ent = json.load(open(tfile)).get(f"{args.preset}_{args.dtype}_{n}")
                if isinstance(ent, dict):
                    traffic, traffic_source = ent.get("bytes"), ent.get("source")
                elif ent is not None:
                    traffic, traffic_source = ent, "profiles/traffic.json (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes)"
            except Exception:
                traffic = None
        if traffic_source:
            traffic_source += "; a constant from that profiling run, not measured by this bench run"
        peak_kernel, peak_torch = hbm_copy_probe(dev)
        peak_measured = max(peak_kernel, peak_torch)
        line = {
            "metric": "env_steps_per_sec", "value": total_steps / dt, "unit": "env-steps/s", "n_gpus": world,
            "steps": K, "warmup": W, "ms_per_step": dt / K * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64" if args.dtype == "f32_state" else args.dtype, "data": "synthetic",  # the arithmetic type
            "config": {"workload": f"{n} parallel arenas per GPU, SimpleDuel3 preset {args.preset} "
                                   f"({p.nr_happy}+{p.nr_grumpy} robots, {p.nb_pos}+{p.nb_neg} balls, "
                                   f"{int(p.arena_w)}x{int(p.arena_h)}), {args.policy}-policy rollout, {na} action(s)/arena, "
                                   f"auto-reset on done, {env.lanes_per_env()} lanes per arena ({64 // env.lanes_per_env()} arena(s) per wavefront)"
                                   + (", arenas at uniformly random episode phases (pre-roll of one whole episode outside the timed region)"
                                      if stagger else ", timed right after a fresh reset")
                                   + (f", BUDGETED step ({args.budget} clocks): {n_not_ready} NOT_READY rows not counted" if args.budget else "")
                                   + (", EXACT-TRIG parity build (double-double sin / cos)" if args.exact_trig else "")
                                   + (", fp32 FAST MODE: state and arithmetic in fp32 -- the 1e-5 parity bar holds on quiet steps only "
                                      "(contact steps: statistical, tests/test_gpu_fp32.py); fp64 is the parity mode" if args.dtype == "f32" else "")
                                   + (", FP32 STATE: the arenas' records in HBM are fp32, a step computes in fp64 between loading a record and "
                                      "writing it back (single steps within 1e-5 of the fp64 reference, contact steps included)" if args.dtype == "f32_state" else ""),
                       "arenas_per_gpu": n, "preset": args.preset, "policy": args.policy, "lanes_per_arena": env.lanes_per_env(),
                       "sharding": f"dp{world} (independent arena shards, returns all-gathered every "
                                   f"{gather_every} steps and once after the loop)" if world > 1 else "single GPU",
                       "steps_per_launch": F, "fault_status_bits_seen": status_bits, "staggered_phases": bool(stagger),
                       "fp64_arithmetic": args.dtype != "f32", "state_dtype": "f64" if args.dtype == "f64" else "f32", "exact_trig": bool(args.exact_trig),
                       "step_budget_clocks": args.budget, "not_ready_fraction": n_not_ready / float(n * K),
                       "collectives": {"all_gather_calls": len(pending), "ranks": world, "backend": rrd.backend_name(),
                                       "bytes_per_rank": 4 * n, "gathered_rows": gathered_rows[-1] if gathered_rows else 0,
                                       "every_steps": gather_every} if world > 1 else None},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                         "peak_measured": peak_measured,
                         "peak_measured_how": "device-to-device copy of 1 GiB, read + write bytes / HIP-event time, this run: the better of the "
                                              "library's one-16-B-element-per-thread copy kernel (rr_probe_hbm_copy) and torch's copy_",
                         "peak_measured_copy_kernel": peak_kernel, "peak_measured_torch_copy": peak_torch,
                         "kernel": "k_step", "kernel_ms": kern_ms, "algorithmic_bytes_per_env_step": bytes_per_step,
                         "record_bytes_per_env": env.state_bytes_per_env(),
                         "note": "latency/VALU-bound by construction: 12 dependent sub-steps of fp64 geometry per "
                                 "149-601 B of state; the HBM fraction is reported because BASELINE.json asks for it"},
        }
        if from_reset is not None:
            line["from_reset"] = from_reset
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(args.preset, args.cpu_seconds)
        print(json.dumps(line), flush=True)
    rrd.barrier()
    env.close()


if __name__ == "__main__":
    main()
