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
    assert abs(a.epsilon - .999997 ** 8) < 1e-12
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


@pytest.mark.gpu
def test_train_loop_end_to_end_on_gpu(tmp_path):
    from roborugby_amd.dqn import train
    ck = str(tmp_path / "ck.pt")
    res = train(num_envs=4096, steps=12, device="cuda:0", checkpoint=ck, log_every=0)
    assert res["env_steps_per_sec"] > 1e4 and res["learn_calls"] == 12
    res2 = train(num_envs=4096, steps=3, device="cuda:0", resume=ck, log_every=0)
    assert res2["steps"] == 3
    with pytest.raises(Exception, match="Game mode"):
        train(num_envs=64, steps=1, preset="G", device="cuda:0", log_every=0)
