"""Batched DQN agent (roborugby_amd/dqn.py): host logic on CPU tensors, end-to-end with the env on the GPU.
Mirrors the reference agent's hyper-parameters and call pattern (Training_DQN_pytorch.py:25-197,250-279,317-377)."""
import pytest
import torch

from roborugby_amd.dqn import BatchedDQNAgent, DeepQNetwork


def test_network_and_hyperparameters_match_reference():
    a = BatchedDQNAgent(device="cpu")
    assert [tuple(p.shape) for p in a.Q_eval.parameters()] == [(256, 11), (256,), (256, 256), (256,), (8, 256), (8,)]
    assert (a.gamma, a.epsilon, a.eps_end, a.eps_dec, a.batch_size, a.mem_size, a.target_update_freq) == \
        (.99, 1.0, 0.2, .999997, 2500, 500000, 100000)
    assert a.Q_eval.optimizer.param_groups[0]["lr"] == .0005
    assert isinstance(a.Q_eval, DeepQNetwork)


def test_store_learn_epsilon_and_target_sync():
    a = BatchedDQNAgent(device="cpu", batch_size=64, max_mem_size=1000, target_update_freq=300, seed=1)
    g = torch.Generator().manual_seed(0)
    assert a.learn() is None  # not enough memory yet (Training_DQN_pytorch.py:152-155)
    w0 = a.Q_target.fc1.weight.clone()
    for it in range(8):
        s = torch.rand(100, 11, generator=g)
        act = a.choose_action(s)
        assert act.dtype == torch.int32 and act.shape == (100,) and int(act.min()) >= 0 and int(act.max()) <= 7
        r = s[:, 0] - 0.5
        valid = torch.ones(100, dtype=torch.bool)
        valid[:10] = False  # re-placed arenas are not transitions
        a.store_transition(s, act, r, torch.rand(100, 11, generator=g), torch.zeros(100, dtype=torch.bool), valid=valid)
        loss = a.learn()
        assert loss is not None and torch.isfinite(loss)
    assert a.mem_cntr == 8 * 90
    # epsilon decays once per stored TRANSITION (the reference: once per learn() = per transition), not once per call
    assert abs(a.epsilon - .999997 ** (8 * 90)) < 1e-12 and a.updates == 8
    assert a.target_syncs == 2  # 300 and 600 stored transitions crossed
    # the ring wrapped (720 < 1000 no; force wrap) and the target network was synced once 300 transitions were stored
    assert not torch.equal(a.Q_target.fc1.weight, w0)
    a.store_transition(torch.rand(400, 11), torch.zeros(400, dtype=torch.int32), torch.zeros(400), torch.rand(400, 11),
                       torch.ones(400, dtype=torch.bool))
    assert a.mem_cntr == 1120 and bool(a.terminal_memory[:120].all())


def test_learning_reduces_td_error_on_a_bandit():
    a = BatchedDQNAgent(device="cpu", batch_size=256, max_mem_size=4096, target_update_freq=10 ** 9, seed=2, lr=.005)
    g = torch.Generator().manual_seed(3)
    s = torch.rand(4096, 11, generator=g)
    act = torch.randint(0, 8, (4096,), generator=g)
    r = (act == 3).float()  # action 3 pays 1, terminal
    a.store_transition(s, act, r, s, torch.ones(4096, dtype=torch.bool))
    first = float(a.learn())
    for _ in range(300):
        last = float(a.learn())
    assert last < 0.2 * first
    greedy = a.choose_action(s[:256], epsilon_override=1e-9)
    assert float((greedy == 3).float().mean()) > 0.95


def test_checkpoint_roundtrip():
    a = BatchedDQNAgent(device="cpu", batch_size=8, max_mem_size=64)
    b = BatchedDQNAgent(device="cpu", batch_size=8, max_mem_size=64, seed=5)
    a.epsilon = 0.5
    b.load_state_dict(a.state_dict(), lr_override=1e-5, eps_dec_override=.9999995)
    assert b.epsilon == 0.5 and b.eps_dec == .9999995 and b.Q_eval.optimizer.param_groups[0]["lr"] == 1e-5
    assert torch.equal(a.Q_eval.fc3.weight, b.Q_eval.fc3.weight)


def test_epsilon_schedule_is_per_transition_at_any_batch_width():
    """5.4e5 transitions take epsilon from 1.0 to the 0.2 floor (Training_DQN_pytorch.py:186-191 with eps_dec .999997),
    whether they arrive one per call or 65,536 per call."""
    import math
    need = math.log(0.2) / math.log(.999997)
    for n in (1000, 65536):
        a = BatchedDQNAgent(device="cpu", batch_size=8, max_mem_size=2 * n)
        s, z = torch.zeros(n, 11), torch.zeros(n)
        calls = 0
        while a.epsilon > 0.2:
            a.store_transition(s, z.int(), z, s, z.bool())
            a.learn()
            calls += 1
        assert abs(calls - need / n) <= 1.0, (n, calls, need / n)


def test_resume_syncs_the_target_within_one_period():
    """after load_state_dict the sync cadence is anchored to the live transition counter, not the old run's."""
    a = BatchedDQNAgent(device="cpu", batch_size=8, max_mem_size=4096, target_update_freq=1000)
    s, z = torch.rand(900, 11), torch.zeros(900)
    for _ in range(5):  # 4,500 transitions: next sync due at 5,000 in THAT run
        a.store_transition(s, z.int(), z, s, z.bool())
        a.learn()
    assert a._next_target_sync == 5000
    b = BatchedDQNAgent(device="cpu", batch_size=8, max_mem_size=4096, target_update_freq=1000, seed=9)
    b.load_state_dict(a.state_dict())
    assert b.mem_cntr == 0 and b._next_target_sync == 1000
    w = b.Q_target.fc1.weight.clone()
    for _ in range(2):  # 1,800 transitions after the resume: one period has passed -> synced
        b.store_transition(s, z.int(), torch.rand(900), s, z.bool())
        b.learn()
    assert b.target_syncs == a.target_syncs + 1 and not torch.equal(b.Q_target.fc1.weight, w)


@pytest.mark.gpu
def test_train_loop_end_to_end_on_gpu(tmp_path):
    from roborugby_amd.dqn import train
    ck = str(tmp_path / "ck.pt")
    res = train(num_envs=4096, steps=12, device="cuda:0", checkpoint=ck, log_every=0)
    # (overlap mode: the k updates of an iteration run next to that iteration's env.step, on the replay as it stood before it,
    # so the first iteration has nothing to learn from)
    assert res["env_steps_per_sec"] > 1e4 and res["overlap_learn"] and res["learn_calls"] == (12 - 1) * 4
    res2 = train(num_envs=4096, steps=3, device="cuda:0", resume=ck, log_every=0)
    assert res2["steps"] == 3
    res3 = train(num_envs=4096, steps=12, device="cuda:0", log_every=0, overlap_learn=False)  # the serial order is still there
    assert not res3["overlap_learn"] and res3["learn_calls"] == 12 * 4
    with pytest.raises(Exception, match="Game mode"):
        train(num_envs=64, steps=1, preset="G", device="cuda:0", log_every=0)


@pytest.mark.gpu
def test_config5_trains_at_65536_arenas(tmp_path):
    """BASELINE config 5 at its stated size: 65,536 arenas of preset T driving the batched DQN loop with learning on --
    replay filling, k gradient steps per vector step (HIP graph once the memory is full), per-transition schedules -- plus
    a checkpoint whose resume continues the episodes instead of restarting them."""
    import roborugby_amd as rr
    from roborugby_amd.dqn import train
    ck = str(tmp_path / "ck65536.pt")
    res = train(num_envs=65536, steps=40, device="cuda:0", checkpoint=ck, log_every=0, replay_vector_steps=8)
    assert res["num_envs"] == 65536 and res["learn_calls"] == (40 - 1) * 4 and res["batch_size"] == 32768
    assert res["replay_transitions"] == 8 * 65536 and res["epsilon"] == 0.2  # 2.6 M transitions: the floor (5.4e5 suffice)
    assert res["target_syncs"] == 0 and res["target_update_freq"] == 64 * 65536
    assert res["env_steps_per_sec"] > 2e6
    saved = torch.load(ck, map_location="cuda:0")
    step, ints = saved["env_state"]["step"], saved["episode"]["ints"]
    assert torch.equal(step, ints[:, 1])                      # steps in the running episode travel with the checkpoint
    assert float((step == 40).float().mean()) > 0.99          # (the few arenas whose spawn faulted were re-placed on the way)
    assert int(ints[:, 0].min()) >= 1                          # episode index: keys the reset RNG after the resume
    res2 = train(num_envs=65536, steps=5, device="cuda:0", resume=ck, log_every=0, replay_vector_steps=8)
    assert res2["steps"] == 5


@pytest.mark.gpu
def test_config5_learns_to_beat_the_random_policy():
    """Two episodes' worth of vector steps at 65,536 arenas (39 M env-steps, a few seconds): the greedy policy's mean episode
    return on a separate evaluation batch is far above the random policy's (profiles/r02/dqn_T_65536.json holds the long curve:
    random 119, greedy 4,003 after one episode, ~30,000 after ten; the reference's own bar is "Avg score 1800 @ ~450 games",
    Training_DQN_pytorch.py:240-249)."""
    from roborugby_amd.dqn import train
    res = train(num_envs=65536, steps=600, device="cuda:0", log_every=0, eval_every=300, eval_envs=8192)
    curve = res["curve"]
    assert len(curve) == 3 and curve[0]["greedy_return"] is None
    rand = curve[0]["random_return"]
    best = max(c["greedy_return"] for c in curve[1:])
    print(f"random policy {rand:.0f}, greedy after 300 / 600 vector steps {curve[1]['greedy_return']:.0f} / {curve[2]['greedy_return']:.0f}; "
          f"training {res['env_steps_per_sec'] / 1e6:.1f} M env-steps/s, greedy rollout {res['rollout_env_steps_per_sec_greedy_policy'] / 1e6:.1f} M")
    assert abs(rand) < 1000 and best > rand + 1500
    assert res["epsilon"] == 0.2 and res["learn_calls"] == 599 * 4


@pytest.mark.gpu
def test_config5_learns_under_the_budgeted_step_and_the_pytorch_path_still_trains():
    """the same loop with the env's budgeted step on: arenas whose step is still in progress report NOT_READY, their rows are no
    transitions, and the transition they complete later carries the action they accepted -- the agent still learns; and the PyTorch
    learn path (fused=False: autograd + torch.optim.Adam, what the fused kernels are tested against) still runs the loop."""
    from roborugby_amd.dqn import train
    res = train(num_envs=65536, steps=600, device="cuda:0", log_every=0, eval_every=600, eval_envs=8192, step_budget_clocks=100_000)
    rand, best = res["curve"][0]["random_return"], res["curve"][-1]["greedy_return"]
    print(f"budgeted step: random {rand:.0f}, greedy after 600 vector steps {best:.0f}; training {res['env_steps_per_sec'] / 1e6:.1f} M env-steps/s")
    assert res["fused_learn_step"] and best > rand + 1500
    res2 = train(num_envs=16384, steps=30, device="cuda:0", log_every=0, fused=False)
    assert not res2["fused_learn_step"] and res2["learn_calls"] == 29 * 4


def test_wide_batch_linear_has_the_gradients_of_a_plain_linear():
    """the learn step's split weight-gradient GEMM (dqn._WideBatchLinear): same forward, same gradients up to summation order"""
    import torch
    from roborugby_amd import dqn
    torch.manual_seed(0)
    net_a = dqn.DeepQNetwork(5e-4, 11, 256, 256, 8).double()
    net_b = dqn.DeepQNetwork(5e-4, 11, 256, 256, 8).double()
    net_b.load_state_dict(net_a.state_dict())
    x = torch.randn(8192, 11, dtype=torch.float64)
    act = torch.randint(0, 8, (8192, 1))
    tgt = torch.randn(8192, dtype=torch.float64)

    def loss_of(net, plain):
        if plain:
            h = torch.relu(net.fc1(x)); h = torch.relu(net.fc2(h)); q = net.fc3(h)
        else:
            q = dqn._linear(net.fc3, torch.relu(dqn._linear(net.fc2, torch.relu(dqn._linear(net.fc1, x)))))
        return ((q.gather(1, act).squeeze(1) - tgt) ** 2).mean()
    la, lb = loss_of(net_a, True), loss_of(net_b, False)
    assert torch.equal(la, lb)
    la.backward(); lb.backward()
    for (n, pa), (_, pb) in zip(net_a.named_parameters(), net_b.named_parameters()):
        assert torch.allclose(pa.grad, pb.grad, rtol=1e-10, atol=1e-13), n
    assert dqn._linear(net_a.fc1, x[:100]).shape == (100, 256)  # narrow batches take the plain layer
