"""DeepQN policy kernels (K2) through the C ABI: bit-exact vs the oracle, and vs the reference's own logits (fixture)."""
import hashlib

import numpy as np
import pytest
import torch

from coevonet_amd import deepqn as dq
from oracle import ref_port as rp
from tests.util import load_golden, sha

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tiled", [False, True])
@pytest.mark.parametrize("C,n_actions,rows", [(4, 6, [3, 1, 10]), (6, 18, [16, 2]), (3, 6, [2, 5]), (5, 18, [4])])
def test_dqn_forward_bit_exact_vs_oracle(C, n_actions, rows, tiled):
    """both fc1 layouts of the slab (streamed: v_mfma_f32_4x4x1, lane = output; tiled: v_mfma_f32_16x16x4 without operand
    moves - include/coevo.h COEVO_DQN_FC1_TILED): the same sequential-k chains, the oracle's bits"""
    torch.manual_seed(C * 10 + n_actions)
    nets = []
    for _ in rows:
        flat, shapes = rp.dqn_init(C, n_actions)
        nets.append(rp.dqn_mutate_torch(flat, shapes, 0.02))
    g = np.random.Generator(np.random.PCG64(7))
    frames = [g.integers(0, 256, size=(r, 84, 84, C), dtype=np.uint8) for r in rows]
    frames[0][0, :, :, :] = 0          # a constant frame: zero variance in conv1's BatchNorm statistics
    logits, actions = dq.batched_actions(nets, frames, C, n_actions, fc1_tiled=tiled)
    row = 0
    for net, fr in zip(nets, frames):
        for r in range(fr.shape[0]):
            a, want = rp.dqn_forward(net, C, n_actions, fr[r])
            assert np.array_equal(logits[row].view(np.uint32), want.view(np.uint32)), (row, logits[row], want)
            assert actions[row] == a
            row += 1


def test_dqn_matches_reference_logits_fixture():
    """the product's DeepQN mirror on the widened pin (six nets: C = 3, 4, 5, 6 planes, 6 / 18 actions; eight frames each incl.
    all-0 / all-255 / constant planes): same parameters as the reference's (sha256), logits within fp32 summation noise of the
    reference's, and - the equality that guards the build-defined BatchNorm order - HIP == oracle bit for bit on these frames"""
    from tests.util import DQN_FRAME_KINDS, dqn_golden_frames
    for case in load_golden("deepqn_forward.json")["cases"]:
        torch.manual_seed(case["torch_seed"])
        C, n = case["C"], case["n_actions"]
        net = dq.DeepQN(C, n, "float32")
        for p in net.parameters():
            p.data += torch.normal(0, case["mutate_std"], size=p.size())
        assert sha(net.flat()) == case["weights_sha256"]
        frames = dqn_golden_frames(C, case["frame_pcg_seed"])
        assert hashlib.sha256(frames.tobytes()).hexdigest() == case["frame_sha256"]
        logits, actions = dq.batched_actions([net.flat()], [frames], C, n)   # all eight frames of the net in one launch
        for r in range(len(DQN_FRAME_KINDS)):
            x = torch.from_numpy(frames[r]).to(torch.float32).permute(2, 0, 1).unsqueeze(0)  # preprocess_observation
            out = net.forward(x).numpy()[0]
            ref = np.array(case["logits"][r], dtype=np.float32)
            np.testing.assert_allclose(out, ref, rtol=1e-4, atol=2e-5)
            srt = np.sort(ref)[::-1]
            if srt[0] - srt[1] > 1e-3:
                assert net.determine_action(x, None) == int(np.argmax(ref))
            a, want = rp.dqn_forward(net.flat(), C, n, frames[r])
            assert np.array_equal(logits[r][:n].view(np.uint32), want.view(np.uint32)), (C, n, DQN_FRAME_KINDS[r])
            assert np.array_equal(out.view(np.uint32), want.view(np.uint32)) and actions[r] == a


def test_dqn_nan_propagates_and_no_action_is_an_error():
    """a NaN weight in conv2 travels through BatchNorm, ReLU (NaN kept, as torch.relu does) and both Linear layers: every
    logit is NaN (checked on the oracle), and the reference's determine_action would return -1 there (Atari/deepqn.py:55-62:
    no `>` comparison holds) - this surface raises ValueError instead of acting on it.  A healthy net is unaffected"""
    C, n = 4, 6
    torch.manual_seed(5)
    flat, shapes = rp.dqn_init(C, n)
    good = rp.dqn_mutate_torch(flat, shapes, 0.02)
    bad = good.copy()
    bad[int(np.prod(shapes[0])) + 32 + 100] = np.nan          # one conv2 weight
    g = np.random.Generator(np.random.PCG64(3))
    frames = [g.integers(0, 256, size=(2, 84, 84, C), dtype=np.uint8) for _ in range(2)]
    a, want = rp.dqn_forward(bad, C, n, frames[1][0])
    assert a == -1 and np.isnan(want).all()
    with pytest.raises(ValueError):
        dq.batched_actions([good, bad], frames, C, n)
    logits, actions = dq.batched_actions([good, good], frames, C, n)
    a0, w0 = rp.dqn_forward(good, C, n, frames[1][1])
    assert actions[3] == a0 and np.array_equal(logits[3].view(np.uint32), w0.view(np.uint32))
