"""debug aid: DeepQN forward vs the oracle; with a DQ_DUMP build, conv1's raw / normalised output of row 0 vs torch"""
import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coevonet_amd import deepqn as dq, lib as L
from oracle import ref_port as rp
torch.manual_seed(3)
C, n = 4, 6
dump = int(os.environ.get("DQ_DUMP", "0"))
nets = []
for _ in range(2):
    flat, shapes = rp.dqn_init(C, n); nets.append(rp.dqn_mutate_torch(flat, shapes, 0.02))
g = np.random.Generator(np.random.PCG64(7))
frames = [g.integers(0, 256, size=(r, 84, 84, C), dtype=np.uint8) for r in (8, 8)]
if not dump:
    frames[0][1] = 0
    logits, actions = dq.batched_actions(nets, frames, C, n)
    row = 0
    for net, fr in zip(nets, frames):
        for r in range(fr.shape[0]):
            a, want = rp.dqn_forward(net, C, n, fr[r])
            print(row, "ok" if np.array_equal(logits[row].view(np.uint32), want.view(np.uint32)) else ("BAD", logits[row][:3], want[:3]))
            row += 1
    sys.exit(0)
dev = "cuda"
stride = int(L.load().coevo_dqn_slab_stride(C, n))
flat = torch.from_numpy(np.stack(nets).astype(np.float32)).to(dev)
slab = torch.zeros(2, stride, dtype=torch.float32, device=dev)
L.call("coevo_dqn_pack", L._p(flat), L._p(slab), 2, C, n)
tasks = np.zeros(2, dtype=L.DQN_TASK_DTYPE); tasks[0] = (0, 0, 8); tasks[1] = (stride, 8, 8)
fr = torch.from_numpy(np.concatenate(frames)).to(dev)
d_tasks = L.tasks_to_device(tasks, dev)
actions = torch.zeros(16, dtype=torch.int32, device=dev); status = torch.zeros(1, dtype=torch.int32, device=dev)
ws = torch.zeros(int(L.load().coevo_dqn_workspace_bytes(16)) // 4, dtype=torch.float32, device=dev)
L.call("coevo_dqn_forward_argmax", L._p(slab), L._p(d_tasks), 2, 8, 16, C, n, L._p(fr), L._p(actions), None, L._p(status), L._p(ws))
torch.cuda.synchronize()
got = ws[:32 * 401].cpu().numpy().reshape(32, 401)[:, :400]
f = torch.from_numpy(nets[0])
w1 = f[:32 * C * 64].reshape(32, C, 8, 8); b1 = f[32 * C * 64:32 * C * 64 + 32]
x = torch.from_numpy(frames[0][0].astype(np.float32) / 255.0).permute(2, 0, 1)[None]
y = torch.nn.functional.conv2d(x, w1, b1, stride=4)[0].reshape(32, 400)
if dump == 1:
    P = int(L.load().coevo_dqn_param_count(C, n))
    g1 = f[P - 320:P - 288]; be1 = f[P - 288:P - 256]
    m = y.mean(1, keepdim=True); v = y.var(1, unbiased=False, keepdim=True)
    y = torch.relu((y - m) / torch.sqrt(v + 1e-5) * g1[:, None] + be1[:, None])
err = np.abs(got - y.numpy())
print("max err", err.max(), "bad entries", int((err > 1e-3).sum()), "of", err.size)
bad = np.argwhere(err > 1e-3)
if len(bad):
    print("bad channels", sorted(set(bad[:, 0].tolist())))
    print("bad positions", sorted(set(bad[:, 1].tolist()))[:80])
