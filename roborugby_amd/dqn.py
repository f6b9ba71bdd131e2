"""Batched DQN trainer: the vectorised counterpart of the reference's Training_DQN_pytorch.py (BASELINE.json config 5).

Same agent as the reference (`DeepQNetwork` Training_DQN_pytorch.py:25-67: Linear 11->256->256->8 with ReLU, Adam
lr 5e-4, MSE; `DQNAgent` :70-197: gamma .99, epsilon-greedy, target network, uniform replay) and the same call pattern as
its main loop (:317-377: choose_action -> env.step -> store_transition (+ the grumpy team's transition when it has robots)
-> learn()), but every tensor -- observations, replay memory, networks -- stays on the MI355X and one call handles all N
arenas.

What "one learn() per env.step" becomes when a step delivers N transitions (the reference: N = 1).  Every schedule the
reference ties to its step counter is kept PER TRANSITION, so it means the same at any N:
  * epsilon:  eps <- max(eps * eps_dec ** n_new, eps_end) for the n_new transitions stored since the last update
              (reference: one factor eps_dec per learn() = per transition, :186-191; 1.0 -> 0.2 after 5.4e5 transitions);
  * target:   Q_target <- Q_eval whenever `target_update_freq` more transitions have been stored (reference :187; in the
              reference that is also 100,000 gradient updates -- its target network is nearly frozen -- so train() scales
              it with the batch: max(100,000, target_sync_vector_steps * N) transitions);
  * replay:   `replay_vector_steps` vector steps deep (at least the reference's 500,000 transitions);
  * update-to-data: the reference draws 2,500 samples per transition, which at 65,536 transitions per step would be
              1.6e8 samples per step.  Here `updates_per_step` (k) gradient steps of `batch_size` (B) samples follow every
              vector step: k * B / N samples per transition (defaults k = 4, B = 32,768: 2 per transition at 65,536 arenas).

    python -m roborugby_amd.dqn --num-envs 65536 --steps 3000 --eval-every 300 --out profiles/r02/dqn_T_65536.json
"""
import argparse
import copy
import ctypes as C
import json
import math
import os
import time

import torch
import torch.nn as nn
import torch.nn.functional as F


class _WideBatchLinear(torch.autograd.Function):
    """y = x W^T + b for a batch far wider than the layer (32,768 samples against 8..256 features).  The forward is F.linear.
    In the backward the weight gradient g^T x is a GEMM with a tiny output and a 32,768-long reduction, which the GEMM library
    runs on a handful of workgroups (139 / 65 / 63 us for the three layers on an MI355X -- 40 % of a learn() step): here the
    batch is cut into slices, one batched GEMM forms the partial products on more of the chip and a small reduction adds them
    (config 5: 14.7 -> 20.4 M env-steps/s over 2,000 vector steps)."""
    SLICE = int(os.environ.get("RR_DQN_SLICE", "1024"))  # samples per slice (0: plain layers); 512 / 1024 / 2048 / 4096 measured: 18.0 / 20.4 / 20.3 / 19.5 M env-steps/s

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        return F.linear(x, w, b)

    @staticmethod
    def backward(ctx, g):
        x, w = ctx.saved_tensors
        gx = g @ w if ctx.needs_input_grad[0] else None
        S = _WideBatchLinear.SLICE
        if S <= 0 or x.shape[0] % S:  # (_linear only routes whole multiples here; anything else takes the plain GEMM)
            return gx, g.t() @ x, g.sum(0)
        n = x.shape[0] // S
        gw = torch.bmm(g.reshape(n, S, -1).transpose(1, 2), x.reshape(n, S, -1)).sum(0)  # reshape: g may arrive non-contiguous
        return gx, gw, g.sum(0)


def _linear(layer, x):
    if _WideBatchLinear.SLICE > 0 and x.dim() == 2 and x.shape[0] >= 8 * _WideBatchLinear.SLICE and x.shape[0] % _WideBatchLinear.SLICE == 0 and torch.is_grad_enabled():
        return _WideBatchLinear.apply(x, layer.weight, layer.bias)
    return layer(x)


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
        x = F.relu(_linear(self.fc1, state.float()))
        x = F.relu(_linear(self.fc2, x))
        return _linear(self.fc3, x)


class BatchedDQNAgent:
    """DQNAgent (Training_DQN_pytorch.py:70-197) with [N]-batched choose_action/store_transition and device replay."""
    PERMUTE_LIMIT = 1 << 20  # sample without replacement (reference: np.random.choice(replace=False)) while a permutation is cheap

    def __init__(self, gamma=.99, epsilon=1.0, lr=.0005, input_dims=11, batch_size=2500, n_actions=8,
                 max_mem_size=500000, eps_end=0.2, eps_dec=.999997, fc1_dims=256, fc2_dims=256,
                 target_update_freq=100000, device="cpu", seed=0, use_graph=True, fused=None):
        self.gamma, self.epsilon, self.eps_end, self.eps_dec = gamma, epsilon, eps_end, eps_dec
        self.n_actions, self.mem_size, self.batch_size = n_actions, int(max_mem_size), int(batch_size)
        self.target_update_freq = int(target_update_freq)
        self.device = torch.device(device)
        self.mem_cntr = 0            # transitions stored so far (the reference's mem_cntr)
        self._eps_cntr = 0           # ... of which the epsilon schedule has already been charged
        self._next_target_sync = self.target_update_freq
        self.target_syncs = 0
        self.updates = 0
        self.gen = torch.Generator(device=self.device)
        self.gen.manual_seed(seed)
        self._seed = int(seed)
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
        # The fused learn step (include/roborugby_amd.h: rr_dqn_update; csrc/rr_dqn.hip): forward of both nets, TD target,
        # backward, weight-gradient reduction and Adam in two launches on the fp32 matrix cores -- for exactly this network
        # (11 -> 256 -> 256 -> 8) on the GPU, batch a multiple of 64.  Default: on whenever it applies.  Off: the PyTorch path
        # (autograd + torch.optim.Adam), which is also the reference the fused step is tested against.
        can_fuse = (self.device.type == "cuda" and (input_dims, fc1_dims, fc2_dims, n_actions) == (11, 256, 256, 8)
                    and self.batch_size % 64 == 0)
        self.fused = can_fuse if fused is None else bool(fused)
        if self.fused and not can_fuse:
            raise ValueError("fused=True needs a GPU, the reference's 11-256-256-8 network and a batch that is a multiple of 64")
        self._fused_h = None
        self._lr, self._betas, self._adam_eps = lr, (0.9, 0.999), 1e-8
        if self.fused:
            from . import _lib
            self._rrlib = _lib.load()
            h = C.c_void_p()
            _lib.check_dqn(self._rrlib.rr_dqn_create(self.device.index or 0, C.byref(h)), "rr_dqn_create", self._rrlib)
            self._fused_h = h
            self._fused_loss = torch.zeros(1, device=self.device)
        self._act_calls = 0  # choose_action calls so far: keys the fused kernel's epsilon draws (checkpointed)

    def close(self):
        """rr_dqn_destroy: the handle holds the per-workgroup partial gradients and the Adam moments (~73 MB of device memory)."""
        h, self._fused_h = getattr(self, "_fused_h", None), None
        if h:
            self._rrlib.rr_dqn_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @torch.no_grad()
    def choose_action(self, observation, epsilon_override=None):
        """[N,11] -> int32 [N]: epsilon-greedy per arena (Training_DQN_pytorch.py:138-149)."""
        eps = self.epsilon if epsilon_override is None else epsilon_override
        n = observation.shape[0]
        if self.fused and n % 64 == 0 and observation.dtype == torch.float32 and observation.shape[1] == 11:
            return self._act_fused(observation, eps)  # one launch: tiled forward on the matrix cores + argmax + epsilon draw
        greedy = self.Q_eval(observation).argmax(dim=1)
        if eps <= 0:
            return greedy.to(torch.int32)
        rand = torch.randint(0, self.n_actions, (n,), generator=self.gen, device=self.device)
        explore = torch.rand(n, generator=self.gen, device=self.device) <= eps
        return torch.where(explore, rand, greedy).to(torch.int32)

    @torch.no_grad()
    def store_transition(self, state, action, reward, state_, done, valid=None):
        """Appends N transitions to the ring (Training_DQN_pytorch.py:126-136); rows with valid=False are skipped
        (the dummy transition of an arena that was only re-placed by auto-reset)."""
        if (self.fused and state.dtype == torch.float32 and state_.dtype == torch.float32 and reward.dtype == torch.float32
                and state.shape[0] <= self.mem_size):
            return self._store_fused(state, action, reward, state_, done, valid)
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

    def _store_fused(self, state, action, reward, state_, done, valid):
        """rr_dqn_store: the compaction of the valid rows and the five ring writes in one launch"""
        from . import _lib
        n = state.shape[0]
        s, s2, r = state.contiguous(), state_.contiguous(), reward.contiguous()
        a = action.to(torch.int32).contiguous()
        d = done.to(torch.bool).contiguous()
        v = valid.to(torch.bool).contiguous() if valid is not None else None
        if not hasattr(self, "_store_count"):
            self._store_count = torch.zeros(1, dtype=torch.int32, device=self.device)
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None  # noqa: E731
        _lib.check_dqn(self._rrlib.rr_dqn_store(self._fused_h, p(s), p(a), p(r), p(s2), p(d), p(v), n, self.mem_cntr, self.mem_size,
                                            p(self.state_memory), p(self.new_state_memory), p(self.action_memory), p(self.reward_memory),
                                            p(self.terminal_memory), p(self._store_count),
                                            C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)), "rr_dqn_store", self._rrlib)
        self.mem_cntr += n if v is None else int(self._store_count.item())  # (the one host read of the call, as in the PyTorch path)

    def _sample(self, max_mem):
        if max_mem <= self.PERMUTE_LIMIT:
            return torch.randperm(max_mem, generator=self.gen, device=self.device)[:self.batch_size]  # replace=False
        # a multi-million-entry memory: independent draws (two of B draws coincide with probability ~B^2 / 2 max_mem per batch)
        return torch.randint(0, max_mem, (self.batch_size,), generator=self.gen, device=self.device)

    def _act_fused(self, observation, eps):
        from . import _lib
        obs = observation.contiguous()
        n = obs.shape[0]
        out = torch.empty(n, dtype=torch.int32, device=self.device)
        ptrs = (C.c_void_p * 6)(*[p.data_ptr() for p in self.Q_eval.parameters()])
        self._act_calls += 1
        _lib.check_dqn(self._rrlib.rr_dqn_act(self._fused_h, C.byref(ptrs), C.c_void_p(obs.data_ptr()), n, float(min(max(eps, 0.0), 1.0)),
                                          int(self._seed), self._act_calls & 0xFFFFFFFF, C.c_void_p(out.data_ptr()), None,
                                          C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)), "rr_dqn_act", self._rrlib)
        self._keep_obs = obs  # read asynchronously
        return out

    def _fused_args(self, batch):
        from . import _lib
        a = _lib.RRDqnArgs()
        a.struct_size, a.batch = C.sizeof(_lib.RRDqnArgs), int(batch.shape[0])
        for k, (pe, pt) in enumerate(zip(self.Q_eval.parameters(), self.Q_target.parameters())):
            assert pe.is_contiguous() and pt.is_contiguous() and pe.dtype == torch.float32
            a.eval_params[k], a.target_params[k] = pe.data_ptr(), pt.data_ptr()
        a.state_memory, a.new_state_memory = self.state_memory.data_ptr(), self.new_state_memory.data_ptr()
        a.action_memory, a.reward_memory = self.action_memory.data_ptr(), self.reward_memory.data_ptr()
        a.terminal_memory, a.batch_index = self.terminal_memory.data_ptr(), batch.data_ptr()
        a.gamma, a.lr = self.gamma, self._lr
        a.beta1, a.beta2, a.eps = self._betas[0], self._betas[1], self._adam_eps
        a.loss_out = self._fused_loss.data_ptr()
        return a

    def _learn_fused(self, max_mem):
        """sampling in PyTorch (one or two kernels), everything else in rr_dqn_update's two launches"""
        from . import _lib
        batch = self._sample(max_mem).contiguous()
        args = self._fused_args(batch)
        _lib.check_dqn(self._rrlib.rr_dqn_update(self._fused_h, C.byref(args), C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)),
                   "rr_dqn_update", self._rrlib)
        self._keep = batch  # the launch reads it asynchronously
        return self._fused_loss[0]

    def fused_grads(self, batch):
        """gradient of the loss on the given replay rows as rr_dqn_grads computes it, as a dict keyed like Q_eval's parameters"""
        from . import _lib
        n = self._rrlib.rr_dqn_param_count()
        flat = torch.empty(n, device=self.device)
        args = self._fused_args(batch.contiguous())
        _lib.check_dqn(self._rrlib.rr_dqn_grads(self._fused_h, C.byref(args), C.c_void_p(flat.data_ptr()),
                                            C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)), "rr_dqn_grads", self._rrlib)
        torch.cuda.current_stream(self.device).synchronize()
        names = ["fc2.weight", "fc1.weight", "fc3.weight", "fc1.bias", "fc2.bias", "fc3.bias"]  # the handle's flat order
        shapes = dict(self.Q_eval.named_parameters())
        out, off = {}, 0
        for nm in names:
            k = shapes[nm].numel()
            out[nm] = flat[off:off + k].view_as(shapes[nm]).clone()
            off += k
        return out, float(self._fused_loss[0])

    def _learn_core(self, max_mem):
        """sampling + TD target + one Adam step (Training_DQN_pytorch.py:156-186); everything on the device, no host sync"""
        self.Q_eval.optimizer.zero_grad(set_to_none=False)
        batch = self._sample(max_mem)
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
        small kernels and launch-bound).  Any failure leaves the eager path in place.  The warm-up iterations PyTorch wants
        before a capture are REAL Adam steps on a side stream; afterwards the parameters, the optimizer state (moments, step
        count) and the sampling generator are restored from the copies taken on entry, so a graph run makes exactly the updates
        -- and the random draws -- an eager run makes."""
        self._graph_tried = True
        opt = self.Q_eval.optimizer
        saved_opt = copy.deepcopy(opt.state_dict())
        saved_gen = self.gen.get_state()
        saved_params = [p.detach().clone() for p in self.Q_eval.parameters()]
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
            with torch.no_grad():  # undo the warm-up: parameters, Adam moments / step count, generator
                for p, q in zip(self.Q_eval.parameters(), saved_params):
                    p.copy_(q)
            opt.load_state_dict(saved_opt)
            self.gen.set_state(saved_gen)
            with torch.cuda.graph(g):
                self._graph_loss = self._learn_core(max_mem)
            self._graph, self._graph_mem = g, max_mem
        except Exception as ex:  # noqa: BLE001 -- eager is always available
            self._graph = None
            with torch.no_grad():
                for p, q in zip(self.Q_eval.parameters(), saved_params):
                    p.copy_(q)
            opt.load_state_dict(saved_opt)
            self.gen.set_state(saved_gen)
            print(f"[dqn] learn() stays eager (graph capture failed: {type(ex).__name__}: {ex})", flush=True)

    def learn(self):
        """One gradient step (Training_DQN_pytorch.py:151-191) + the per-transition schedules (module docstring)."""
        if self.mem_cntr < self.batch_size:
            return None
        max_mem = min(self.mem_size, self.mem_cntr)
        if self.fused:
            loss = self._learn_fused(max_mem)
        elif self.use_graph and self.device.type == "cuda" and max_mem == self.mem_size and not getattr(self, "_graph_tried", False):
            self._try_capture(max_mem)
        if self.fused:
            pass
        elif getattr(self, "_graph", None) is not None and self._graph_mem == max_mem:
            self._graph.replay()
            loss = self._graph_loss
        else:
            loss = self._learn_core(max_mem)
        self.updates += 1
        # the reference syncs when mem_cntr hits a multiple of target_update_freq; with N transitions per call the
        # counter jumps, so sync whenever a multiple has been crossed
        if self.mem_cntr >= self._next_target_sync:
            with torch.no_grad():  # in place: a captured graph keeps reading these very tensors
                for pt, pe in zip(self.Q_target.parameters(), self.Q_eval.parameters()):
                    pt.copy_(pe)
            self._next_target_sync = (self.mem_cntr // self.target_update_freq + 1) * self.target_update_freq
            self.target_syncs += 1
        # one factor eps_dec per transition stored since the schedule was last charged (reference: per learn() = per transition)
        n_new = self.mem_cntr - self._eps_cntr
        if n_new > 0:
            self.epsilon = max(self.epsilon * math.pow(self.eps_dec, n_new), self.eps_end)
            self._eps_cntr = self.mem_cntr
        self.last_loss = loss
        return self.last_loss

    # whole-agent checkpoint like the reference's pickle (Training_DQN_pytorch.py:373-376), without the replay
    def _fused_adam(self, state=None):
        """the fused step's Adam moments + step count out of (state=None) or into the handle"""
        from . import _lib
        n = self._rrlib.rr_dqn_param_count()
        if state is None:
            m1, m2 = torch.empty(n, device=self.device), torch.empty(n, device=self.device)
            step = C.c_int64(0)
        else:
            m1, m2 = state["exp_avg"].to(self.device).contiguous(), state["exp_avg_sq"].to(self.device).contiguous()
            step = C.c_int64(int(state["step"]))
        st = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        _lib.check_dqn(self._rrlib.rr_dqn_adam_state(self._fused_h, C.c_void_p(m1.data_ptr()), C.c_void_p(m2.data_ptr()), C.byref(step),
                                                 0 if state is None else 1, st), "rr_dqn_adam_state", self._rrlib)
        torch.cuda.current_stream(self.device).synchronize()
        return dict(exp_avg=m1, exp_avg_sq=m2, step=step.value)

    # flat layout of the fused step's Adam moments (csrc/rr_dqn.hip: OFF_W2, OFF_W1, OFF_W3, OFF_B1, OFF_B2, OFF_B3) as indices
    # into Q_eval.parameters() = fc1.weight, fc1.bias, fc2.weight, fc2.bias, fc3.weight, fc3.bias
    _FUSED_ORDER = (2, 0, 4, 1, 3, 5)

    def _flat_from_torch_adam(self, opt_sd):
        """torch.optim.Adam state_dict -> the fused layout; None if the optimizer was never stepped"""
        st = opt_sd.get("state", {})
        if len(st) < 6:
            return None
        params = list(self.Q_eval.parameters())
        cat = lambda key: torch.cat([st[k][key].detach().to(self.device, torch.float32).reshape(-1) for k in self._FUSED_ORDER])
        step = int(torch.as_tensor(st[0]["step"]).item())
        assert sum(params[k].numel() for k in self._FUSED_ORDER) == cat("exp_avg").numel()
        return dict(exp_avg=cat("exp_avg"), exp_avg_sq=cat("exp_avg_sq"), step=step)

    def _torch_adam_from_flat(self, flat):
        """the fused layout -> the `state` part of a torch.optim.Adam state_dict for Q_eval's parameters"""
        params = list(self.Q_eval.parameters())
        state, off = {}, 0
        for k in self._FUSED_ORDER:
            n = params[k].numel()
            state[k] = dict(step=torch.tensor(float(flat["step"]), device=self.device),
                            exp_avg=flat["exp_avg"][off:off + n].reshape(params[k].shape).clone(),
                            exp_avg_sq=flat["exp_avg_sq"][off:off + n].reshape(params[k].shape).clone())
            off += n
        return state

    def state_dict(self):
        """Both optimizer representations travel, so a checkpoint written by the fused agent resumes under --no-fused and the
        other way round: `optimizer` (torch.optim.Adam's) and `fused_adam` (flat moments + step) describe the same Adam state."""
        opt = self.Q_eval.optimizer.state_dict()
        sd = dict(q_eval=self.Q_eval.state_dict(), q_target=self.Q_target.state_dict(), epsilon=self.epsilon, mem_cntr=self.mem_cntr,
                  next_target_sync=self._next_target_sync, updates=self.updates, target_syncs=self.target_syncs,
                  act_calls=self._act_calls, fused=self.fused)
        if self.fused:
            sd["fused_adam"] = self._fused_adam()
            if sd["fused_adam"]["step"] > 0:  # the torch optimizer is never stepped in fused mode: write the moments in its format too
                opt = dict(opt, state=self._torch_adam_from_flat(sd["fused_adam"]))
        else:
            flat = self._flat_from_torch_adam(opt)
            if flat is not None:
                sd["fused_adam"] = flat
        sd["optimizer"] = opt
        return sd

    def load_state_dict(self, sd, lr_override=0.0, epsilon_override=0.0, eps_dec_override=0.0):
        self.Q_eval.load_state_dict(sd["q_eval"])
        self.Q_target.load_state_dict(sd["q_target"])
        self.Q_eval.optimizer.load_state_dict(sd["optimizer"])
        if self.fused:
            flat = sd.get("fused_adam") or self._flat_from_torch_adam(sd["optimizer"])
            if flat is not None:
                self._fused_adam(flat)
            elif sd.get("updates", 0) > 0:
                import warnings
                warnings.warn("checkpoint carries no Adam moments for the fused learn step: Adam restarts from zero moments")
        self._act_calls = int(sd.get("act_calls", 0))
        self.epsilon = sd["epsilon"]
        self.updates, self.target_syncs = sd.get("updates", 0), sd.get("target_syncs", 0)
        # The replay memory is not checkpointed (the reference pickles it with the agent), so the transition counter
        # restarts at what this process has stored: the sync cadence is re-anchored to THAT counter -- a sync within
        # target_update_freq transitions of the resume, as in an uninterrupted run -- never to the old run's total.
        self._eps_cntr = self.mem_cntr
        self._next_target_sync = (self.mem_cntr // self.target_update_freq + 1) * self.target_update_freq
        if lr_override > 0:  # Training_DQN_pytorch.py:297-303
            self._lr = lr_override
            for g in self.Q_eval.optimizer.param_groups:
                g["lr"] = lr_override
        if epsilon_override > 0:
            self.epsilon = epsilon_override
        if eps_dec_override > 0:
            self.eps_dec = eps_dec_override


@torch.no_grad()
def evaluate(env, policy, episodes=1):
    """Mean return per episode of `policy(obs) -> int32 [N]` over every arena of `env` (auto-reset env, time_limit rule):
    resets, then plays `episodes` full episodes; steps that only re-place an arena carry reward 0."""
    obs = env.reset()
    total = torch.zeros(env.num_envs, device=env.device, dtype=torch.float64)
    T = env.spec.max_episode_steps
    n_steps = episodes * T + (episodes - 1)  # one re-placing call between consecutive episodes
    for _ in range(n_steps):
        obs, reward, done, info = env.step(policy(obs).view(-1, 1))
        total += reward.double()
    return float(total.mean() / episodes), float(total.std() / episodes)


def train(num_envs=65536, steps=300, preset="T", device="cuda:0", seed=0, checkpoint=None, resume=None,
          log_every=50, learn=True, mem_size=None, dtype="f64", updates_per_step=4, batch_size=None,
          replay_vector_steps=32, target_sync_vector_steps=64, eps_dec=.999997, eps_end=0.2, eval_every=0,
          eval_envs=16384, out=None, overlap_learn=True, step_budget_clocks=0, fused=None):
    """The main loop of Training_DQN_pytorch.py:317-377 over a batched env.  Returns a dict of throughput / score /
    the return curve (greedy policy vs the random policy on a separate evaluation batch, every `eval_every` steps).

    overlap_learn: the k gradient steps that follow a vector step run on a second HIP stream WHILE the simulator steps the
    next one (under a contact-seeking policy `k_step` is one long launch bound by its slowest arenas and leaves most CUs idle).
    Stream-ordered, deterministic: choose_action(t) -> [side: learn x k on the replay up to t-1] || [main: env.step(t)] ->
    store_transition(t) waits for the side stream -> choose_action(t+1).  The updates see the replay one vector step later
    than in the serial order (same number of updates, same schedules)."""
    import roborugby_amd as rr
    # step_budget_clocks > 0: the budgeted step (include/roborugby_amd.h) -- an arena whose step is still in progress reports
    # NOT_READY; its row is no transition yet, and the transition it completes later carries the action it ACCEPTED
    env = rr.make("RoboRugbySimpleDuel-v3", num_envs=num_envs, preset=preset, device=device, seed=seed, dtype=dtype,
                  step_budget_clocks=step_budget_clocks)
    p = env.preset
    if p.game_mode:  # Training_DQN_pytorch.py:233-234
        raise Exception("Game mode settings are enabled in RR_Constants.")
    n_teams = 2 if p.nr_grumpy > 0 else 1
    B = int(batch_size or min(32768, max(env.spec.max_episode_steps * 8 + 100, num_envs // 2)))
    agent = BatchedDQNAgent(input_dims=env.observation_space.shape[0], batch_size=B, n_actions=env.action_space.n, device=device,
                            seed=seed, max_mem_size=mem_size or max(500000, replay_vector_steps * num_envs * n_teams),
                            target_update_freq=max(100000, target_sync_vector_steps * num_envs * n_teams),
                            eps_dec=eps_dec, eps_end=eps_end, fused=fused)
    if resume:
        ck = torch.load(resume, map_location=device)
        agent.load_state_dict(ck["agent"])
        env.load_checkpoint_state(ck)  # state, episode bookkeeping (the episode index keys the reset RNG), parity build: the scratch rect
        observation = env.get_game_state()
    else:
        observation = env.reset()
    grumpy = p.nr_grumpy > 0
    obs_grumpy = env.get_game_state(int_team=-1) if grumpy else None
    eval_env, curve = None, []
    if eval_every:
        eval_env = rr.make("RoboRugbySimpleDuel-v3", num_envs=eval_envs, preset=preset, device=device, seed=seed + 1000003, dtype=dtype)
        gen = torch.Generator(device=device)
        gen.manual_seed(seed + 7)
        rand_ret, rand_std = evaluate(eval_env, lambda o: torch.randint(0, 8, (o.shape[0],), generator=gen, device=o.device, dtype=torch.int32))
        curve.append(dict(env_steps=0, vector_steps=0, greedy_return=None, random_return=rand_ret, epsilon=agent.epsilon))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    t_eval = 0.0
    score_sum = torch.zeros(num_envs, device=device)
    real_rows = torch.zeros((), dtype=torch.int64, device=device)  # rows that were transitions (not re-placements, not NOT_READY)
    overlap = bool(overlap_learn and learn and torch.device(device).type == "cuda")
    main_stream = torch.cuda.current_stream(torch.device(device))
    side = torch.cuda.Stream(device=torch.device(device)) if overlap else None
    parked = None  # budgeted step: arenas whose step was still in progress after the last call
    for i in range(steps):
        action = agent.choose_action(observation)
        if grumpy:
            action_grumpy = agent.choose_action(obs_grumpy)
        if parked is not None:  # a parked arena ignores the new action: the transition it will complete belongs to the one it accepted
            action = torch.where(parked, accepted, action)
            if grumpy:
                action_grumpy = torch.where(parked, accepted_g, action_grumpy)
        acts = action.view(-1, 1)
        learned = None
        if overlap:  # the gradient steps on what the replay holds so far, next to the simulator's step
            chosen = torch.cuda.Event()
            chosen.record(main_stream)
            with torch.cuda.stream(side):
                side.wait_event(chosen)  # choose_action has read Q_eval
                for _ in range(updates_per_step):
                    agent.learn()
                learned = torch.cuda.Event()
                learned.record(side)
        if step_budget_clocks and checkpoint and i == steps - 1:
            # a checkpoint must not catch an arena parked mid-step (rr_set_state on resume would drop the rest of that step and
            # the action it accepted): the last call before it runs synchronously -- parked arenas finish, nobody parks
            env.set_step_budget(0)
        observation_, reward, done, info = env.step(acts)
        real = (info.status & (1024 | 16384)) == 0  # a call that only re-placed the arena, or left its step unfinished, is not a transition
        if step_budget_clocks:
            parked, accepted = (info.status & 16384) != 0, action
            if grumpy:
                accepted_g = action_grumpy
        score_sum += reward
        real_rows += real.sum()
        if learn:
            if learned is not None:
                main_stream.wait_event(learned)  # the sampled rows are read, Q_eval is written: the ring and the net are ours again
            agent.store_transition(observation, action, reward, observation_, done, valid=real)
            if grumpy:
                agent.store_transition(obs_grumpy, action_grumpy, info.dblGrumpyScore, info.adblGrumpyState, done, valid=real)
            if not overlap:
                for _ in range(updates_per_step):
                    agent.learn()
        observation = observation_
        obs_grumpy = info.adblGrumpyState
        if log_every and (i + 1) % log_every == 0:
            torch.cuda.synchronize()
            lr_, _, ll, cnt = env.episode_stats()
            fin = cnt > 0
            avg = float(lr_[fin].mean()) if bool(fin.any()) else float("nan")
            print(f"step {i + 1} epsilon {agent.epsilon:.6f} updates {agent.updates} target-syncs {agent.target_syncs} finished-episodes "
                  f"{int(cnt.sum())} avg-last-return {avg:.1f} loss {float(agent.last_loss) if agent.last_loss is not None else float('nan'):.4f}",
                  flush=True)
        if eval_every and (i + 1) % eval_every == 0:
            torch.cuda.synchronize()
            te = time.perf_counter()
            g_ret, g_std = evaluate(eval_env, lambda o: agent.choose_action(o, epsilon_override=0.0))
            curve.append(dict(env_steps=(i + 1) * num_envs, vector_steps=i + 1, greedy_return=g_ret, greedy_return_std=g_std,
                              random_return=rand_ret, epsilon=agent.epsilon, updates=agent.updates, target_syncs=agent.target_syncs))
            print(f"[eval] after {(i + 1) * num_envs:,} env-steps: greedy return {g_ret:.1f} (random policy {rand_ret:.1f})", flush=True)
            torch.cuda.synchronize()
            t_eval += time.perf_counter() - te
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0 - t_eval
    if checkpoint:
        os.makedirs(os.path.dirname(os.path.abspath(checkpoint)), exist_ok=True)
        torch.save(dict(agent=agent.state_dict(), **env.checkpoint_state()), checkpoint)
    lr_, _, ll, cnt = env.episode_stats()
    n_real = int(real_rows.item())
    # env steps = the rows that stepped an arena (auto-reset calls count, as in bench.py's random line they are subtracted only there;
    # NOT_READY rows of the budgeted step never count)
    env_steps = num_envs * steps if not step_budget_clocks else n_real
    res = dict(env_steps_per_sec=env_steps / dt, seconds=dt, num_envs=num_envs, steps=steps, transitions=n_real,
               step_budget_clocks=step_budget_clocks,
               learn_calls=agent.updates, updates_per_step=updates_per_step if learn else 0, batch_size=B,
               samples_per_transition=(updates_per_step * B / (num_envs * n_teams)) if learn else 0.0, overlap_learn=overlap,
               replay_transitions=agent.mem_size, target_update_freq=agent.target_update_freq, target_syncs=agent.target_syncs,
               epsilon=agent.epsilon, eps_dec=eps_dec, eps_end=eps_end, finished_episodes=int(cnt.sum()), fused_learn_step=bool(agent.fused),
               mean_step_reward=float(score_sum.mean() / steps), preset=preset, dtype=dtype, curve=curve)
    if eval_every:
        # throughput of the rollout alone under the policy training converged to (greedy, contact-seeking), next to the random policy's
        for name, pol in (("greedy", lambda o: agent.choose_action(o, epsilon_override=0.0)),
                          ("random", lambda o: torch.randint(0, 8, (o.shape[0],), device=o.device, dtype=torch.int32))):
            o = env.reset()
            torch.cuda.synchronize()
            ts = time.perf_counter()
            for _ in range(200):
                o, _, _, _ = env.step(pol(o).view(-1, 1))
            torch.cuda.synchronize()
            res[f"rollout_env_steps_per_sec_{name}_policy"] = num_envs * 200 / (time.perf_counter() - ts)
        eval_env.close()
    env.close()
    if out:
        os.makedirs(os.path.dirname(os.path.abspath(out)) or ".", exist_ok=True)
        with open(out, "w") as f:
            json.dump(res, f, indent=1)
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
    ap.add_argument("--updates-per-step", type=int, default=4)
    ap.add_argument("--batch-size", type=int, default=None)
    ap.add_argument("--eps-dec", type=float, default=.999997)
    ap.add_argument("--eval-every", type=int, default=0)
    ap.add_argument("--eval-envs", type=int, default=16384)
    ap.add_argument("--log-every", type=int, default=50)
    ap.add_argument("--out", default=None)
    ap.add_argument("--no-overlap", action="store_true", help="gradient steps after the simulator's step instead of next to it")
    ap.add_argument("--no-fused", action="store_true", help="learn step through PyTorch autograd + torch.optim.Adam instead of rr_dqn_update")
    ap.add_argument("--budget", type=int, default=0, help="step_budget_clocks of the env (the budgeted step; 0 = synchronous)")
    a = ap.parse_args()
    res = train(a.num_envs, a.steps, a.preset, a.device, a.seed, a.checkpoint, a.resume, learn=not a.no_learn,
                updates_per_step=a.updates_per_step, batch_size=a.batch_size, eps_dec=a.eps_dec, eval_every=a.eval_every,
                eval_envs=a.eval_envs, log_every=a.log_every, out=a.out, overlap_learn=not a.no_overlap,
                step_budget_clocks=a.budget, fused=False if a.no_fused else None)
    print(json.dumps({k: v for k, v in res.items() if k != "curve"}))


if __name__ == "__main__":
    main()
