"""Closed-form test functions shared by the golden generator and the tests (TEST INFRASTRUCTURE).

Everything here is integer arithmetic or a single correctly-rounded IEEE operation, so the same
definitions in oracle/az_oracle.c and in the HIP engine (alphazero_amd/csrc) agree bit for bit:
  * splitmix64 / board hash / fake policy-value net with dyadic priors,
  * closed-form Dirichlet noise ("hash" noise mode),
  * Philox4x32-10 + the 53-bit uniform used for move sampling,
  * closed-form network weights (no torch RNG, no 4 MB weight fixture).
"""
import numpy as np

M64 = (1 << 64) - 1


def splitmix64(z):
    z = (z + 0x9E3779B97F4A7C15) & M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M64
    return z ^ (z >> 31)


def board_hash(grid, player):
    """hash of the canonical board player*grid, cells row-major"""
    h = 0x9E3779B97F4A7C15
    for v in np.asarray(grid).reshape(-1):
        cg = int(player) * int(v) + 1
        h = ((h ^ cg) * 0x100000001B3) & M64
    return splitmix64(h)


def fakenet(grid, player, action_size):
    """returns (probs float32[A], v_net float) -- canonical frame, like PolicyValueNetwork.predict"""
    h = board_hash(grid, player)
    probs = np.zeros(action_size, dtype=np.float32)
    for a in range(action_size):
        w = 1 + (splitmix64((h + (a + 1) * 0x9E3779B97F4A7C15) & M64) >> 58)
        probs[a] = np.float32(w) / np.float32(4096.0)
    t = splitmix64(h ^ 0xD1B54A32D192ED03)
    sel = (t >> 10) & 15
    v = (np.float32(int(t & 1023)) - np.float32(512.0)) / np.float32(512.0)
    if sel == 0:
        v = np.float32(0.0)
    if sel == 1:
        v = np.float32(6.103515625e-05)
    return probs, float(v)


def hash_noise(grid, player, actions):
    """closed-form Dirichlet replacement: eta[a] = w_a / sum(w) for the root's child actions"""
    h = board_hash(grid, player)
    w = [1 + (splitmix64((h + (a + 1) * 0xBF58476D1CE4E5B9) & M64) >> 54) for a in actions]
    tot = sum(w)
    return {a: float(wi) / float(tot) for a, wi in zip(actions, w)}


def philox4x32(k0, k1, c0, c1, c2, c3):
    m32 = 0xFFFFFFFF
    for _ in range(10):
        p0 = 0xD2511F53 * c0
        p1 = 0xCD9E8D57 * c2
        n0 = ((p1 >> 32) ^ c1 ^ k0) & m32
        n1 = p1 & m32
        n2 = ((p0 >> 32) ^ c3 ^ k1) & m32
        n3 = p0 & m32
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + 0x9E3779B9) & m32
        k1 = (k1 + 0xBB67AE85) & m32
    return c0, c1, c2, c3


P_TIE_SELECT, P_NOISE_NORMAL, P_NOISE_BOOST, P_MOVE_SAMPLE, P_TIE_MOVE, P_ROLLOUT_EXPAND, P_PLAYOUT = 1, 2, 3, 4, 5, 6, 7


def u53(a, b):
    return (float(a >> 5) * 67108864.0 + float(b >> 6)) * (1.0 / 9007199254740992.0)


def move_sample_u(seed, game_id, ply):
    r = philox4x32(seed, game_id, ply, 0xFFFF, P_MOVE_SAMPLE, 0)
    return u53(r[0], r[1])


# ---- action encoding -------------------------------------------------------------------------

def move_to_action(game, move, n=None):
    if game == "othello":
        return n * n if tuple(move) == (n, n) else int(move[0]) * n + int(move[1])
    if game == "tictactoe":
        return 3 * int(move[0]) + int(move[1])
    return int(move)


def action_to_move(game, action, n=None):
    if game == "othello":
        return (n, n) if action == n * n else (action // n, action % n)
    if game == "tictactoe":
        return (action // 3, action % 3)
    return int(action)


# ---- closed-form network weights ---------------------------------------------------------------

def _name_seed(name):
    h = 0xCBF29CE484222325
    for ch in name.encode():
        h = ((h ^ ch) * 0x100000001B3) & M64
    return h


def _uniform_pm1(seed, n):
    """n values in [-1, 1), exact multiples of 2^-23"""
    out = np.empty(n, dtype=np.float64)
    for i in range(n):
        out[i] = float(splitmix64((seed + i) & M64) >> 40) / 8388608.0 - 1.0
    return out


def _uniform_pm1_fast(seed, n):
    i = np.arange(n, dtype=np.uint64)
    z = (np.uint64(seed) + i)
    with np.errstate(over="ignore"):
        z = z + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return (z >> np.uint64(40)).astype(np.float64) / 8388608.0 - 1.0


def closed_form_state_dict(shapes):
    """shapes: {state_dict key: shape}.  Returns {key: float32 ndarray} with non-trivial BN statistics."""
    out = {}
    for name, shape in shapes.items():
        n = int(np.prod(shape)) if len(shape) else 1
        if name.endswith("num_batches_tracked"):
            out[name] = np.zeros(shape, dtype=np.int64)
            continue
        u = _uniform_pm1_fast(_name_seed(name), n)
        if name.endswith("running_var"):
            val = 1.0 + 0.5 * u          # [0.5, 1.5)
        elif name.endswith("running_mean"):
            val = 0.1 * u
        elif ("bn" in name) and name.endswith("weight"):
            val = 1.0 + 0.25 * u         # gamma
        elif ("bn" in name) and name.endswith("bias"):
            val = 0.1 * u                # beta
        elif name.endswith("weight"):
            fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else 1
            val = u * (2.0 / np.sqrt(fan_in))
        else:                            # linear / conv bias
            val = 0.1 * u
        out[name] = val.astype(np.float32).reshape(shape)
    return out
