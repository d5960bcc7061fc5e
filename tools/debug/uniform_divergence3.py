import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import oracle as O
from alphazero_amd import engine as E
from alphazero_amd.games.othello import OthelloNet
def srt(d):
    meta = d["meta"].cpu().numpy(); order = np.lexsort((meta[:, 1], meta[:, 0]))
    return {k: v.cpu().numpy()[order] for k, v in d.items()}
for bias in (0.0, 60.0):
    torch.manual_seed(4)
    net = OthelloNet(n=8).eval()
    with torch.no_grad():
        net.fc_probs.bias[27] += bias
    sd = {k: v.detach().cpu().numpy() for k, v in net.state_dict().items() if not k.endswith("num_batches_tracked")}
    onet = O.ConvNet(0, 8, 8, sd)
    hnet = net.to_hip(max_batch=16)
    grids, players, _ = O.random_positions(0, 8, 8, 5, 3, 100)
    canon = (grids * players[:, None]).astype(np.float32)
    for B in (1, 4, 5, 6, 7, 8, 12):
        p, v = hnet.forward(torch.as_tensor(canon[:B], device="cuda"))
        op, ov = onet.forward(canon[:B])
        print("bias", bias, "forward B", B, "probs eq", np.array_equal(p.cpu().numpy(), op), "v eq", np.array_equal(v.cpu().numpy(), ov), v.cpu().numpy()[:8], ov[:8])
    for slots in (4, 5, 6, 7, 8):
        ref = O.selfplay(0, 8, 8, slots, 30, ("conv", onet), seed=8)
        eng = E.SelfPlayEngine(0, 8, 8, n_slots=slots, n_sim=30, net=hnet, seed=8)
        got = srt(eng.run(slots))
        bad = np.flatnonzero((got["visits"] != ref["visits"]).any(1))
        print("bias", bias, "slots", slots, "bad samples", len(bad), "games", sorted(set(got["meta"][bad, 0])) if len(bad) else [])
