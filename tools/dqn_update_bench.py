"""Times the fused DQN learn step (rr_dqn_update: two launches) and the fused choose_action (rr_dqn_act) alone on the chip:
HIP-event time per call, and the learn step's 18.3 GFLOP (B = 32,768; forward of both nets, backward, weight gradients) against the
157 TFLOP/s dense fp32 matrix-core peak (MI355X_MICROARCH.md).  RR_LIB_PATH selects an alternative build.
usage: python tools/dqn_update_bench.py [batch] [iters]"""
import ctypes as C, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from roborugby_amd import _lib
from roborugby_amd.dqn import BatchedDQNAgent
B = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 100
ag = BatchedDQNAgent(batch_size=B, device="cuda:0", seed=0, max_mem_size=2 ** 21)
g = torch.Generator(device="cuda:0").manual_seed(1)
ag.state_memory.copy_(torch.rand(ag.mem_size, 11, generator=g, device="cuda:0") * 300)
ag.new_state_memory.copy_(ag.state_memory + 1)
ag.action_memory.copy_(torch.randint(0, 8, (ag.mem_size,), generator=g, device="cuda:0"))
ag.reward_memory.normal_(generator=g)
ag.mem_cntr = ag.mem_size
idx = torch.randint(0, ag.mem_size, (B,), generator=g, device="cuda:0")
args = ag._fused_args(idx)
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for _ in range(5):
    _lib.check(ag._rrlib.rr_dqn_update(ag._fused_h, C.byref(args), st), "rr_dqn_update")
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    _lib.check(ag._rrlib.rr_dqn_update(ag._fused_h, C.byref(args), st), "rr_dqn_update")
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / iters
flop = 2 * B * (2 * (11 * 256 + 256 * 256 + 256 * 8) + (256 * 8 + 256 * 256) + (11 * 256 + 256 * 256 + 256 * 8))
obs = torch.rand(65536, 11, generator=g, device="cuda:0") * 300
for _ in range(3):
    ag.choose_action(obs)
e0.record()
for _ in range(iters):
    ag.choose_action(obs)
e1.record(); torch.cuda.synchronize()
us_act = e0.elapsed_time(e1) * 1e3 / iters
fl_act = 2 * 65536 * (11 * 256 + 256 * 256 + 256 * 8)
print(f"{os.environ.get('RR_LIB_PATH', 'product library')}: rr_dqn_update B={B}: {us:.1f} us/call = {flop / us / 1e6:.1f} TFLOP/s "
      f"({100 * flop / us / 1e6 / 157:.1f} % of the 157 TFLOP/s fp32 matrix-core peak); rr_dqn_act 65,536 rows: {us_act:.1f} us = {fl_act / us_act / 1e6:.1f} TFLOP/s")
