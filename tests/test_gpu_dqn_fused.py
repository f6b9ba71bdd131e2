"""The fused DQN learn step (rr_dqn_update / rr_dqn_grads, roborugby_amd/csrc/rr_dqn.hip) against PyTorch autograd + torch.optim.Adam
on the same replay rows: the reference's learn() (Training_DQN_pytorch.py:151-191) for its network (:25-67).
Both fp32 paths sum up to 32,768 products per gradient entry in different orders, so the yardstick is autograd in fp64: the fused
gradient must be within 1e-5 of each parameter gradient's scale (max |g|) of it -- or, where fp32 autograd itself is further away
than that (the widest batch), no further than fp32 autograd is (x 1.25)."""
import copy

import numpy as np
import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu


def _agent(batch, seed=0, mem=100_000, **kw):
    from roborugby_amd.dqn import BatchedDQNAgent
    ag = BatchedDQNAgent(batch_size=batch, device="cuda:0", seed=seed, max_mem_size=mem, **kw)
    g = torch.Generator(device="cuda:0").manual_seed(100 + seed)
    n = mem
    # observations shaped like the env's (angles / distances of a few hundred), rewards of a few units, 3 % terminal rows
    s = torch.rand(n, 11, generator=g, device="cuda:0") * 360.0 - 50.0
    s2 = s + torch.randn(n, 11, generator=g, device="cuda:0") * 3.0
    ag.state_memory.copy_(s); ag.new_state_memory.copy_(s2)
    ag.action_memory.copy_(torch.randint(0, 8, (n,), generator=g, device="cuda:0"))
    ag.reward_memory.copy_(torch.randn(n, generator=g, device="cuda:0") * 2.0)
    ag.terminal_memory.copy_(torch.rand(n, generator=g, device="cuda:0") < 0.03)
    ag.mem_cntr = n
    with torch.no_grad():  # a target net that differs from the eval net, as it does between syncs
        for p in ag.Q_target.parameters():
            p.add_(torch.randn(p.shape, generator=g, device="cuda:0") * 0.01)
    return ag, g


def _torch_grads(ag, batch, dtype=torch.float32):
    net, tgt = copy.deepcopy(ag.Q_eval).to(dtype), copy.deepcopy(ag.Q_target).to(dtype)

    def fwd(n, x):  # DeepQNetwork.forward without its .float()
        return n.fc3(torch.relu(n.fc2(torch.relu(n.fc1(x)))))
    q_eval = fwd(net, ag.state_memory[batch].to(dtype)).gather(1, ag.action_memory[batch].view(-1, 1)).squeeze(1)
    with torch.no_grad():
        q_next = fwd(tgt, ag.new_state_memory[batch].to(dtype)).masked_fill(ag.terminal_memory[batch].view(-1, 1), 0.0)
        y = ag.reward_memory[batch].to(dtype) + ag.gamma * q_next.max(dim=1)[0]
    loss = torch.nn.functional.mse_loss(q_eval, y)
    loss.backward()
    return {k: p.grad.clone() for k, p in net.named_parameters()}, float(loss.detach())


def _rows_off_the_relu_knife_edge(ag, g, batch, tau=1e-3):
    """replay rows none of whose 512 hidden pre-activations (eval net, fp64) lies within tau of 0: there relu'(z) -- 0 or 1 --
    depends on the last bits of z, and a single sample whose mask flips moves a gradient entry by a whole term (observed at
    B = 32,768: one unit of 16.8 M pre-activations, 1e-5 of the gradient's scale), whatever fp32 implementation computes z"""
    cand = torch.randint(0, ag.mem_size, (batch + batch // 8 + 64,), generator=g, device="cuda:0")
    net = copy.deepcopy(ag.Q_eval).double()
    with torch.no_grad():
        z1 = net.fc1(ag.state_memory[cand].double())
        z2 = net.fc2(torch.relu(z1))
    ok = (z1.abs().min(dim=1)[0] > tau) & (z2.abs().min(dim=1)[0] > tau)
    rows = cand[ok][:batch]
    assert rows.numel() == batch
    return rows.contiguous()


@pytest.mark.parametrize("batch", [64, 4096, 32768])
def test_fused_gradient_matches_autograd(batch):
    ag, g = _agent(batch)
    assert ag.fused
    idx = _rows_off_the_relu_knife_edge(ag, g, batch)
    ref, ref_loss = _torch_grads(ag, idx, torch.float64)
    t32, _ = _torch_grads(ag, idx, torch.float32)
    got, loss = ag.fused_grads(idx)
    assert abs(loss - ref_loss) <= 1e-5 * max(1.0, abs(ref_loss)), (loss, ref_loss)
    for k in ref:
        scale = float(ref[k].abs().max())
        err = float((got[k].double() - ref[k]).abs().max())
        err32 = float((t32[k].double() - ref[k]).abs().max())
        print(f"  B={batch} {k}: fused {err / scale:.2e}, fp32 autograd {err32 / scale:.2e} of the gradient's scale (vs fp64 autograd)")
        assert scale > 0 and err <= max(1e-5 * scale, 1.25 * err32), (batch, k, err, err32, scale)


def test_fused_adam_steps_match_torch_adam():
    batch = 8192
    ag, g = _agent(batch, seed=3)
    ref_net = copy.deepcopy(ag.Q_eval)
    opt = torch.optim.Adam(ref_net.parameters(), lr=ag._lr)
    for step in range(5):
        idx = torch.randint(0, ag.mem_size, (batch,), generator=g, device="cuda:0").contiguous()
        # reference step on the copy
        opt.zero_grad()
        q = ref_net(ag.state_memory[idx]).gather(1, ag.action_memory[idx].view(-1, 1)).squeeze(1)
        with torch.no_grad():
            qn = ag.Q_target(ag.new_state_memory[idx]).masked_fill(ag.terminal_memory[idx].view(-1, 1), 0.0)
            y = ag.reward_memory[idx] + ag.gamma * qn.max(dim=1)[0]
        torch.nn.functional.mse_loss(q, y).backward()
        opt.step()
        # fused step on the agent's own parameters
        import ctypes as C
        from roborugby_amd import _lib
        args = ag._fused_args(idx)
        _lib.check(ag._rrlib.rr_dqn_update(ag._fused_h, C.byref(args), C.c_void_p(torch.cuda.current_stream().cuda_stream)), "rr_dqn_update")
        torch.cuda.synchronize()
    for (k, p), q in zip(ag.Q_eval.named_parameters(), ref_net.parameters()):
        d = (p - q).abs()
        # Adam normalises the gradient: an entry whose gradient is ~0 may step differently, everything else agrees closely
        assert float(d.max()) <= 2.5 * ag._lr * 5, (k, float(d.max()))
        assert float(torch.quantile(d.flatten()[:1_000_000], 0.999)) <= 2e-6, (k, float(torch.quantile(d.flatten(), 0.999)))
    st = ag._fused_adam()
    assert st["step"] == 5 and float(st["exp_avg_sq"].min()) >= 0.0


def test_fused_agent_learns_a_bandit():
    """the contextual bandit of tests/test_dqn_agent.py through the fused step: action 3 pays 1 (terminal), the greedy policy finds it"""
    from roborugby_amd.dqn import BatchedDQNAgent
    ag = BatchedDQNAgent(batch_size=4096, device="cuda:0", seed=1, max_mem_size=65536, lr=1e-3)
    assert ag.fused
    g = torch.Generator(device="cuda:0").manual_seed(5)
    for it in range(60):
        s = torch.rand(4096, 11, generator=g, device="cuda:0")
        a = torch.randint(0, 8, (4096,), generator=g, device="cuda:0")
        r = (a == 3).float()
        ag.store_transition(s, a, r, s, torch.ones(4096, dtype=torch.bool, device="cuda:0"))
        for _ in range(4):
            ag.learn()
    s = torch.rand(2048, 11, generator=g, device="cuda:0")
    greedy = ag.choose_action(s, epsilon_override=0.0)
    assert float((greedy == 3).float().mean()) > 0.95
    assert float(ag.last_loss) < 0.05


def test_fused_choose_action_is_the_networks_argmax_and_explores_at_rate_epsilon():
    from roborugby_amd.dqn import BatchedDQNAgent
    ag = BatchedDQNAgent(batch_size=4096, device="cuda:0", seed=2, max_mem_size=8192)
    g = torch.Generator(device="cuda:0").manual_seed(8)
    obs = torch.rand(65536, 11, generator=g, device="cuda:0") * 400.0 - 100.0
    a = ag.choose_action(obs, epsilon_override=0.0)
    with torch.no_grad():
        q = ag.Q_eval(obs)
    ref = q.argmax(dim=1)
    agree = (a.long() == ref)
    # a different action only where the two fp32 forwards disagree about a near-tie of the top two Q values
    top2 = q.topk(2, dim=1)[0]
    gap = (top2[:, 0] - top2[:, 1])
    assert float(agree.float().mean()) > 0.9995 and float(gap[~agree].max() if (~agree).any() else 0.0) < 1e-3 * float(q.abs().max())
    # epsilon-greedy: the explored fraction and the uniformity of the random actions
    e = ag.choose_action(obs, epsilon_override=0.3)
    changed = (e != a)
    frac = float(changed.float().mean())  # 0.3 * 7/8 of the rows end up with another action
    assert abs(frac - 0.3 * 7 / 8) < 0.01, frac
    e2 = ag.choose_action(obs, epsilon_override=0.3)
    assert not torch.equal(e, e2)  # a fresh call counter: fresh draws
    hist = torch.bincount(ag.choose_action(obs, epsilon_override=1.0).long(), minlength=8).float() / obs.shape[0]
    assert float((hist - 0.125).abs().max()) < 0.01


def test_fused_store_transition_equals_the_pytorch_path():
    from roborugby_amd.dqn import BatchedDQNAgent
    a1 = BatchedDQNAgent(batch_size=4096, device="cuda:0", seed=2, max_mem_size=10_000)
    a2 = BatchedDQNAgent(batch_size=4096, device="cuda:0", seed=2, max_mem_size=10_000, fused=False)
    g = torch.Generator(device="cuda:0").manual_seed(3)
    for it in range(5):  # 5 x 3,000 rows into a 10,000-row ring: wraps around, with 10 % of the rows invalid
        n = 3000
        s = torch.rand(n, 11, generator=g, device="cuda:0"); s2 = torch.rand(n, 11, generator=g, device="cuda:0")
        a = torch.randint(0, 8, (n,), generator=g, device="cuda:0", dtype=torch.int32)
        r = torch.randn(n, generator=g, device="cuda:0"); d = torch.rand(n, generator=g, device="cuda:0") < 0.1
        v = (torch.rand(n, generator=g, device="cuda:0") < 0.9) if it % 2 else None
        a1.store_transition(s, a, r, s2, d, valid=v)
        a2.store_transition(s, a, r, s2, d, valid=v)
        assert a1.mem_cntr == a2.mem_cntr
    for name in ("state_memory", "new_state_memory", "action_memory", "reward_memory", "terminal_memory"):
        assert torch.equal(getattr(a1, name), getattr(a2, name)), name


def test_checkpoint_round_trips_between_fused_and_torch_adam():
    """a checkpoint written by the fused agent must resume under fused=False (and the other way round) with the same Adam
    moments and step count, not with zero moments -- and it carries the epsilon-draw counter.  Three updates in one mode, checkpoint,
    load into the other mode: the moments agree entry by entry, and one more update on the same rows moves both agents alike."""
    import ctypes as C
    from roborugby_amd import _lib
    batch = 4096
    for src_fused in (True, False):
        src, g = _agent(batch, seed=5, fused=src_fused, use_graph=False)
        dst, _ = _agent(batch, seed=6, fused=not src_fused, use_graph=False)
        for _ in range(3):
            src.learn()
        src.choose_action(src.state_memory[:1024])
        torch.cuda.synchronize()
        sd = src.state_dict()
        assert sd["act_calls"] == src._act_calls and "fused_adam" in sd and len(sd["optimizer"]["state"]) == 6
        assert sd["fused_adam"]["step"] == 3 and int(torch.as_tensor(sd["optimizer"]["state"][0]["step"]).item()) == 3
        dst.load_state_dict(sd)
        assert dst._act_calls == src._act_calls
        # same Adam state on both sides, in the destination's own representation
        flat_src = sd["fused_adam"]
        flat_dst = dst._fused_adam() if dst.fused else dst._flat_from_torch_adam(dst.Q_eval.optimizer.state_dict())
        assert flat_dst["step"] == 3
        assert torch.equal(flat_dst["exp_avg"].cpu(), flat_src["exp_avg"].cpu()) and torch.equal(flat_dst["exp_avg_sq"].cpu(), flat_src["exp_avg_sq"].cpu())
        assert float(flat_src["exp_avg_sq"].max()) > 0.0
        for p, q in zip(src.Q_eval.parameters(), dst.Q_eval.parameters()):
            assert torch.equal(p, q)
        # one more step on identical rows (same replay contents, same sampler state) in both modes
        dst.state_memory.copy_(src.state_memory); dst.new_state_memory.copy_(src.new_state_memory); dst.action_memory.copy_(src.action_memory)
        dst.reward_memory.copy_(src.reward_memory); dst.terminal_memory.copy_(src.terminal_memory)
        dst.gen.set_state(src.gen.get_state())
        src.learn(); dst.learn()
        torch.cuda.synchronize()
        for (k, p), q in zip(src.Q_eval.named_parameters(), dst.Q_eval.parameters()):
            d = (p - q).detach().abs()
            assert float(d.max()) <= 2.5 * src._lr, (src_fused, k)  # (an entry with a ~0 gradient may step differently)
            assert float(torch.quantile(d.flatten()[:1_000_000], 0.999)) <= 2e-6, (src_fused, k)
        src.close(); dst.close()
        assert src._fused_h is None and dst._fused_h is None
