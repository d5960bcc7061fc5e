import ctypes as C, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from oracle import oracle as O
from alphazero_amd import engine as E
from alphazero_amd.games.othello import OthelloNet
torch.manual_seed(4)
net = OthelloNet(n=8).eval()
with torch.no_grad():
    net.fc_probs.bias[27] += 60.0
sd = {k: v.detach().cpu().numpy() for k, v in net.state_dict().items() if not k.endswith("num_batches_tracked")}
onet = O.ConvNet(0, 8, 8, sd)
hnet = net.to_hip(max_batch=16)
ref = O.selfplay(0, 8, 8, 6, 30, ("conv", onet), seed=8)
def srt(d):
    meta = d["meta"].cpu().numpy(); order = np.lexsort((meta[:, 1], meta[:, 0]))
    return {k: v.cpu().numpy()[order] for k, v in d.items()}
for trial in range(3):
    eng = E.SelfPlayEngine(0, 8, 8, n_slots=6, n_sim=30, net=hnet, seed=8)
    got = srt(eng.run(6))
    bad = np.flatnonzero((got["visits"] != ref["visits"]).any(1))
    print("run() slots=6 trial", trial, "graphs", os.environ.get("AZ_ENGINE_GRAPHS"), "mismatching samples", len(bad), "first", (got["meta"][bad[0]] if len(bad) else None), "replays", eng.stats()["graph_replays"])
    if len(bad):
        i = bad[0]; print(got["visits"][i][got["visits"][i] > 0], ref["visits"][i][ref["visits"][i] > 0]); print(got["pi"][i][got["pi"][i]>0], ref["pi"][i][ref["pi"][i]>0])
# step-wise with 6 slots
eng = E.SelfPlayEngine(0, 8, 8, n_slots=6, n_sim=30, net=hnet, seed=8)
b0 = O.new_board(0, 8, 8)
eng.set_roots(np.stack([b0.grid_np()] * 6), np.ones(6, np.int8))
boards = [O.new_board(0, 8, 8) for _ in range(6)]
trees = [O.MCT(("conv", onet), alpha=0.03, eps=0.25, tie_mode=1, noise_mode=1, seed=8, game_id=g) for g in range(6)]
done = False
for ply in range(60):
    eng.search(30)
    for g in range(6):
        trees[g].set_ply(ply); trees[g].search(boards[g], 30)
        a, N, Q, P, rn = eng.root_children(g); oa, oN, oQ, oP = trees[g].root_children()
        if not (np.array_equal(N, oN) and np.array_equal(Q, oQ) and np.array_equal(P, oP)):
            print("stepwise slots=6: game", g, "ply", ply); print(N, oN); print(P, oP); print(Q, oQ); done = True
    if done: break
    for g in range(6):
        act, _, _ = trees[g].choose(boards[g], 1.0 if ply <= 4 else 0.0)
        O.lib().orc_play(C.byref(boards[g]), act); trees[g].change_root(act)
    eng.advance()
else:
    print("stepwise slots=6 equal")
