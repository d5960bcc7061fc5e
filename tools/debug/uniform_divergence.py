"""finds the first ply at which engine and oracle root statistics diverge under a uniform-prior network"""
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
for gid in range(6):
    eng = E.SelfPlayEngine(0, 8, 8, n_slots=1, n_sim=30, net=hnet, seed=8)
    b = O.new_board(0, 8, 8)
    eng.set_roots(b.grid_np()[None], np.array([1], np.int8), game_ids=np.array([gid], np.uint32))
    t = O.MCT(("conv", onet), alpha=0.03, eps=0.25, tie_mode=1, noise_mode=1, seed=8, game_id=gid)
    ply = 0
    while not O.lib().orc_is_over(C.byref(b)):
        eng.search(30)
        t.set_ply(ply); t.search(b, 30)
        a, N, Q, P, rn = eng.root_children(0)
        oa, oN, oQ, oP = t.root_children()
        if not (np.array_equal(N, oN) and np.array_equal(Q, oQ) and np.array_equal(P, oP)):
            print("game", gid, "ply", ply, "rootN", rn, t.root_n(), "evals", eng.stats()["net_evals"], t.__dict__.get("x"))
            print(" a ", a, oa); print(" N ", N, oN); print(" P ", P, oP); print(" Q ", Q, oQ)
            print(b.grid_np(), b.player)
            break
        temp = 1.0 if ply <= 4 else 0.0
        act, _, _ = t.choose(b, temp)
        eng.advance()
        O.lib().orc_play(C.byref(b), act); t.change_root(act); ply += 1
        m = eng.samples()["meta"].cpu().numpy()
        if m[-1, 3] != act:
            print("game", gid, "ply", ply - 1, "moves differ", m[-1, 3], act); break
    else:
        print("game", gid, "equal", ply)
