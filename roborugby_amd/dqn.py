"""Batched DQN trainer: the vectorised counterpart of the reference's Training_DQN_pytorch.py (BASELINE.json config 5).

Same agent as the reference (`DeepQNetwork` Training_DQN_pytorch.py:25-67: Linear 11->256->256->8 with ReLU, Adam
lr 5e-4, MSE; `DQNAgent` :70-197: gamma .99, epsilon 1.0 decayed by x0.999997 per learn() down to 0.2, target
network copied every `target_update_freq` stored transitions, uniform replay sampled WITHOUT replacement, batch =
max_episode_steps*8+100) and the same call pattern as its main loop (:317-377: choose_action -> env.step ->
store_transition (+ the grumpy team's transition when it has robots) -> learn() once per step), but every tensor --
observations, replay memory, networks -- stays on the MI355X and one call handles all N arenas.

    python -m roborugby_amd.dqn --num-envs 65536 --steps 300
"""
import argparse
import copy
import json
import os
import time

import torch
import torch.nn as nn
import torch.nn.functional as F


class DeepQNetwork(nn.Module):
    def __init__(self, lr, input_dims, fc1_dims, fc2_dims, n_actions):
        super().__init__()
        self.fc1 = nn.Linear(input_dims, fc1_dims)
        self.fc2 = nn.Linear(fc1_dims, fc2_dims)
        self.fc3 = nn.Linear(fc2_dims, n_actions)
        # (the fused single-kernel Adam step where the parameters live on the GPU: the learn() call is launch-bound)
        self.optimizer = torch.optim.Adam(self.parameters(), lr=lr)
        self._lr = lr
        self.loss = nn.MSELoss()

    def forward(self, state):
        x = F.relu(self.fc1(state.float()))
        x = F.relu(self.fc2(x))
        return self.fc3(x)


class BatchedDQNAgent:
    """DQNAgent (Training_DQN_pytorch.py:70-197) with [N]-batched choose_action/store_transition and device replay."""

    def __init__(self, gamma=.99, epsilon=1.0, lr=.0005, input_dims=11, batch_size=2500, n_actions=8,
                 max_mem_size=500000, eps_end=0.2, eps_dec=.999997, fc1_dims=256, fc2_dims=256,
                 target_update_freq=100000, device="cpu", seed=0, use_graph=True):
        self.gamma, self.epsilon, self.eps_end, self.eps_dec = gamma, epsilon, eps_end, eps_dec
        self.n_actions, self.mem_size, self.batch_size = n_actions, int(max_mem_size), int(batch_size)
        self.target_update_freq = int(target_update_freq)
        self.device = torch.device(device)
        self.mem_cntr = 0
        self._next_target_sync = self.target_update_freq
        self.gen = torch.Generator(device=self.device)
        self.gen.manual_seed(seed)
        torch.manual_seed(seed)
        self.Q_eval = DeepQNetwork(lr, input_dims, fc1_dims, fc2_dims, n_actions).to(self.device)
        if self.device.type == "cuda":
            self.Q_eval.optimizer = torch.optim.Adam(self.Q_eval.parameters(), lr=lr, fused=True, capturable=True)
        self.Q_target = copy.deepcopy(self.Q_eval)
        m, d = self.mem_size, self.device
        self.state_memory = torch.zeros(m, input_dims, dtype=torch.float32, device=d)
        self.new_state_memory = torch.zeros(m, input_dims, dtype=torch.float32, device=d)
        self.action_memory = torch.zeros(m, dtype=torch.int64, device=d)
        self.reward_memory = torch.zeros(m, dtype=torch.float32, device=d)
        self.terminal_memory = torch.zeros(m, dtype=torch.bool, device=d)
        self.last_loss = None
        self.use_graph = bool(use_graph)

    @torch.no_grad()
    def choose_action(self, observation, epsilon_override=None):
        """[N,11] -> int32 [N]: epsilon-greedy per arena (Training_DQN_pytorch.py:138-149)."""
        eps = epsilon_override if epsilon_override else self.epsilon
        n = observation.shape[0]
        greedy = self.Q_eval(observation).argmax(dim=1)
        rand = torch.randint(0, self.n_actions, (n,), generator=self.gen, device=self.device)
        explore = torch.rand(n, generator=self.gen, device=self.device) <= eps
        return torch.where(explore, rand, greedy).to(torch.int32)

    @torch.no_grad()
    def store_transition(self, state, action, reward, state_, done, valid=None):
        """Appends N transitions to the ring (Training_DQN_pytorch.py:126-136); rows with valid=False are skipped
        (the dummy transition of an arena that was only re-placed by auto-reset)."""
        if valid is not None:
            idx = valid.nonzero(as_tuple=True)[0]
            state, action, reward, state_, done = state[idx], action[idx], reward[idx], state_[idx], done[idx]
        n = state.shape[0]
        if n == 0:
            return
        pos = (self.mem_cntr + torch.arange(n, device=self.device)) % self.mem_size
        self.state_memory[pos] = state.float()
        self.new_state_memory[pos] = state_.float()
        self.action_memory[pos] = action.long()
        self.reward_memory[pos] = reward.float()
        self.terminal_memory[pos] = done.bool()
        self.mem_cntr += n

    def _learn_core(self, max_mem):
        """sampling + TD target + one Adam step (Training_DQN_pytorch.py:156-186); everything on the device, no host sync"""
        self.Q_eval.optimizer.zero_grad(set_to_none=False)
        batch = torch.randperm(max_mem, generator=self.gen, device=self.device)[:self.batch_size]  # replace=False
        state_batch = self.state_memory[batch]
        new_state_batch = self.new_state_memory[batch]
        reward_batch = self.reward_memory[batch]
        terminal_batch = self.terminal_memory[batch]
        action_batch = self.action_memory[batch]
        q_eval = self.Q_eval(state_batch).gather(1, action_batch.view(-1, 1)).squeeze(1)
        with torch.no_grad():
            q_next = self.Q_target(new_state_batch)
            q_next = q_next.masked_fill(terminal_batch.view(-1, 1), 0.0)
            q_target = reward_batch + self.gamma * q_next.max(dim=1)[0]
        loss = self.Q_eval.loss(q_target, q_eval)
        loss.backward()
        self.Q_eval.optimizer.step()
        return loss.detach()

    def _try_capture(self, max_mem):
        """Once the replay memory is full the learn step has a fixed shape: capture it into a HIP graph (the step is ~25
        small kernels and launch-bound).  Any failure leaves the eager path in place."""
        self._graph_tried = True
        try:
            g = torch.cuda.CUDAGraph()
            if hasattr(g, "register_generator_state"):
                g.register_generator_state(self.gen)
            side = torch.cuda.Stream(device=self.device)
            side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(side):
                for _ in range(2):
                    self._learn_core(max_mem)
            torch.cuda.current_stream(self.device).wait_stream(side)
            with torch.cuda.graph(g):
                self._graph_loss = self._learn_core(max_mem)
            self._graph, self._graph_mem = g, max_mem
        except Exception as ex:  # noqa: BLE001 -- eager is always available
            self._graph = None
            print(f"[dqn] learn() stays eager (graph capture failed: {type(ex).__name__}: {ex})", flush=True)

    def learn(self):
        """One gradient step (Training_DQN_pytorch.py:151-191)."""
        if self.mem_cntr < self.batch_size:
            return None
        max_mem = min(self.mem_size, self.mem_cntr)
        if self.use_graph and self.device.type == "cuda" and max_mem == self.mem_size and not getattr(self, "_graph_tried", False):
            self._try_capture(max_mem)
        if getattr(self, "_graph", None) is not None and self._graph_mem == max_mem:
            self._graph.replay()
            loss = self._graph_loss
        else:
            loss = self._learn_core(max_mem)
        # the reference syncs when mem_cntr hits a multiple of target_update_freq; with N transitions per call the
        # counter jumps, so sync whenever a multiple has been crossed
        if self.mem_cntr >= self._next_target_sync:
            with torch.no_grad():  # in place: a captured graph keeps reading these very tensors
                for pt, pe in zip(self.Q_target.parameters(), self.Q_eval.parameters()):
                    pt.copy_(pe)
            self._next_target_sync = (self.mem_cntr // self.target_update_freq + 1) * self.target_update_freq
        self.epsilon = max(self.epsilon * self.eps_dec, self.eps_end)
        self.last_loss = loss
        return self.last_loss

    # whole-agent checkpoint like the reference's pickle (Training_DQN_pytorch.py:373-376), without the replay
    def state_dict(self):
        return dict(q_eval=self.Q_eval.state_dict(), q_target=self.Q_target.state_dict(),
                    optimizer=self.Q_eval.optimizer.state_dict(), epsilon=self.epsilon, mem_cntr=self.mem_cntr,
                    next_target_sync=self._next_target_sync)

    def load_state_dict(self, sd, lr_override=0.0, epsilon_override=0.0, eps_dec_override=0.0):
        self.Q_eval.load_state_dict(sd["q_eval"])
        self.Q_target.load_state_dict(sd["q_target"])
        self.Q_eval.optimizer.load_state_dict(sd["optimizer"])
        self.epsilon = sd["epsilon"]
        self._next_target_sync = sd.get("next_target_sync", self.target_update_freq)
        if lr_override > 0:  # Training_DQN_pytorch.py:297-303
            for g in self.Q_eval.optimizer.param_groups:
                g["lr"] = lr_override
        if epsilon_override > 0:
            self.epsilon = epsilon_override
        if eps_dec_override > 0:
            self.eps_dec = eps_dec_override


def train(num_envs=65536, steps=300, preset="T", device="cuda:0", seed=0, checkpoint=None, resume=None,
          log_every=50, learn=True, mem_size=None, dtype="f64"):
    """The main loop of Training_DQN_pytorch.py:317-377 over a batched env.  Returns a dict of throughput/score."""
    import roborugby_amd as rr
    env = rr.make("RoboRugbySimpleDuel-v3", num_envs=num_envs, preset=preset, device=device, seed=seed, dtype=dtype)
    p = env.preset
    if p.game_mode:  # Training_DQN_pytorch.py:233-234
        raise Exception("Game mode settings are enabled in RR_Constants.")
    agent = BatchedDQNAgent(input_dims=env.observation_space.shape[0], batch_size=env.spec.max_episode_steps * 8 + 100,
                            n_actions=env.action_space.n, device=device, seed=seed,
                            max_mem_size=mem_size or max(500000, 8 * num_envs))
    if resume:
        ck = torch.load(resume, map_location=device)
        agent.load_state_dict(ck["agent"])
        st = ck["env_state"]
        env.set_state(st["robots"], st["robots_i"], st["balls"], st["step"])
        observation = env.get_game_state()
    else:
        observation = env.reset()
    grumpy = p.nr_grumpy > 0
    obs_grumpy = env.get_game_state(int_team=-1) if grumpy else None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    score_sum = torch.zeros(num_envs, device=device)
    for i in range(steps):
        action = agent.choose_action(observation)
        acts = action.view(-1, 1)
        if grumpy:
            action_grumpy = agent.choose_action(obs_grumpy)
        observation_, reward, done, info = env.step(acts)
        real = (info.status & 1024) == 0  # a call that only re-placed the arena is not a transition
        score_sum += reward
        if learn:
            agent.store_transition(observation, action, reward, observation_, done, valid=real)
            if grumpy:
                agent.store_transition(obs_grumpy, action_grumpy, info.dblGrumpyScore, info.adblGrumpyState, done, valid=real)
            agent.learn()
        observation = observation_
        obs_grumpy = info.adblGrumpyState
        if log_every and (i + 1) % log_every == 0:
            lr_, _, ll, cnt = env.episode_stats()
            fin = cnt > 0
            avg = float(lr_[fin].mean()) if bool(fin.any()) else float("nan")
            print(f"step {i + 1} epsilon {agent.epsilon:.6f} finished-episodes {int(cnt.sum())} "
                  f"avg-last-return {avg:.1f} loss {float(agent.last_loss) if agent.last_loss is not None else float('nan'):.4f}",
                  flush=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if checkpoint:
        os.makedirs(os.path.dirname(os.path.abspath(checkpoint)), exist_ok=True)
        torch.save(dict(agent=agent.state_dict(), env_state=env.get_state()), checkpoint)
    lr_, _, ll, cnt = env.episode_stats()
    res = dict(env_steps_per_sec=num_envs * steps / dt, seconds=dt, num_envs=num_envs, steps=steps,
               learn_calls=steps if learn else 0, epsilon=agent.epsilon, finished_episodes=int(cnt.sum()),
               mean_step_reward=float(score_sum.mean() / steps))
    env.close()
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--num-envs", type=int, default=65536)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--preset", default="T")
    ap.add_argument("--device", default="cuda:0")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--checkpoint", default=None)
    ap.add_argument("--resume", default=None)
    ap.add_argument("--no-learn", action="store_true")
    a = ap.parse_args()
    res = train(a.num_envs, a.steps, a.preset, a.device, a.seed, a.checkpoint, a.resume, learn=not a.no_learn)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
