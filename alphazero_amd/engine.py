"""Host side of the MI355X self-play engine: thin Python over the C ABI (include/az_amd.h).

torch is used for device memory and streams only; all computation is in libaz_amd.so.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import (EVAL_FAKE, EVAL_NET, EVAL_ROLLOUT, GAME_IDS, NOISE_HASH, NOISE_OFF, NOISE_PHILOX, TIE_LOWEST, TIE_RANDOM,  # noqa: F401
                   EngineCfg, EngineStats, check, lib)


def _stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def game_shape(game, board_size=None, board_width=7, board_height=6):
    """(game id, H, W, action size) for the reference's game names / config fields"""
    if game == "othello":
        n = 6 if board_size is None else board_size  # OthelloConfig.board_size default (othello.py:22)
        return GAME_IDS[game], n, n, n * n + 1
    if game == "connect4":
        return GAME_IDS[game], board_height, board_width, board_width
    if game == "tictactoe":
        return GAME_IDS[game], 3, 3, 9
    raise ValueError(f"unknown game {game!r}")


class _DevView:
    """exposes a raw device pointer through __cuda_array_interface__ so torch can wrap it without a copy"""

    def __init__(self, ptr, shape, typestr, owner):
        self.__cuda_array_interface__ = {"shape": tuple(shape), "typestr": typestr, "data": (int(ptr), False),
                                         "version": 2, "strides": None}
        self._owner = owner


def _wrap(ptr, shape, dtype, owner):
    typestr = {torch.int8: "|i1", torch.float32: "<f4", torch.int32: "<i4", torch.uint8: "|u1"}[dtype]
    if int(np.prod(shape)) == 0:
        return torch.empty(shape, dtype=dtype, device="cuda")
    return torch.as_tensor(_DevView(ptr, shape, typestr, owner), device="cuda")


# ---------------------------------------------------------------------------------------------- rules
def legal_batch(game_id, H, W, grids, players, for_player=None):
    """Board.get_moves for n positions: uint8 [n, A] legality table (device tensors in, device tensor out)."""
    n = players.numel()
    A = H * W + 1 if game_id == 0 else (W if game_id == 1 else 9)
    assert grids.dtype == torch.int8 and players.dtype == torch.int8 and grids.is_cuda and grids.is_contiguous()
    out = torch.empty((n, A), dtype=torch.uint8, device=grids.device)
    fp = for_player.data_ptr() if for_player is not None else None
    check(lib().az_board_legal_batch(game_id, H, W, grids.data_ptr(), players.data_ptr(), fp, n, out.data_ptr(), _stream_ptr()))
    return out


def play_batch(game_id, H, W, grids, players, actions):
    """Board.play_move for n positions; status[i] = 0 or AZ_EILLEGAL (the reference raises ValueError)."""
    n = players.numel()
    assert actions.dtype == torch.int32 and grids.is_contiguous()
    og, op = torch.empty_like(grids), torch.empty_like(players)
    st = torch.empty(n, dtype=torch.int32, device=grids.device)
    check(lib().az_board_play_batch(game_id, H, W, grids.data_ptr(), players.data_ptr(), actions.data_ptr(), n,
                                    og.data_ptr(), op.data_ptr(), st.data_ptr(), _stream_ptr()))
    return og, op, st


def status_batch(game_id, H, W, grids, players):
    """(is_game_over uint8, get_winner int8 [2 where not over], sum(player*grid) int32) for n positions."""
    n = players.numel()
    over = torch.empty(n, dtype=torch.uint8, device=grids.device)
    win = torch.empty(n, dtype=torch.int8, device=grids.device)
    score = torch.empty(n, dtype=torch.int32, device=grids.device)
    check(lib().az_board_status_batch(game_id, H, W, grids.data_ptr(), players.data_ptr(), n, over.data_ptr(),
                                      win.data_ptr(), score.data_ptr(), _stream_ptr()))
    return over, win, score


def augment_samples(game_id, H, W, samples):
    """symmetry twins of the samples with move_idx >= 2, in the reference's order (trainer.py:275-284).
    samples: dict of CUDA tensors (state int8 [S,H,W], pi float32 [S,A], z int8 [S], meta int32 [S,4]).
    Returns the twins as a dict of the same form; meta[:, 3] is the transformation code 1..7."""
    st, pi, z, meta = (samples[k].contiguous() for k in ("state", "pi", "z", "meta"))
    S, A = z.shape[0], pi.shape[1]
    n = C.c_int64()
    check(lib().az_augment_count(game_id, meta.data_ptr() if S else None, S, C.byref(n), _stream_ptr()))
    N = n.value
    out = {"state": torch.empty((N, H, W), dtype=torch.int8, device=st.device), "pi": torch.empty((N, A), dtype=torch.float32, device=st.device),
           "z": torch.empty(N, dtype=torch.int8, device=st.device), "meta": torch.empty((N, 4), dtype=torch.int32, device=st.device)}
    if N:
        check(lib().az_augment(game_id, H, W, st.data_ptr(), pi.data_ptr(), z.data_ptr(), meta.data_ptr(), S, out["state"].data_ptr(),
                               out["pi"].data_ptr(), out["z"].data_ptr(), out["meta"].data_ptr(), N, _stream_ptr()))
    return out


TRANSFORM_NAMES = [None, "reflection_horizontal", "rotation_90", "reflection_horizontal+rotation_90", "rotation_180",
                   "reflection_horizontal+rotation_180", "rotation_270", "reflection_horizontal+rotation_270"]


# ---------------------------------------------------------------------------------------------- network
class HipNet:
    """Device policy-value network built from a reference state_dict (eval-mode BN folded at upload)."""

    def __init__(self, game_id, H, W, state_dict, max_batch=4096):
        self.game_id, self.H, self.W, self.max_batch = game_id, H, W, max_batch
        h = C.c_void_p()
        check(lib().az_net_create(game_id, H, W, max_batch, C.byref(h)))
        self.h = h
        self.A = lib().az_net_action_size(self.h)
        self.load_state_dict(state_dict)

    def load_state_dict(self, state_dict):
        """weights in (update_network, trainer.py:383-387).  CUDA tensors never leave the device: BatchNorm is folded and
        the weights are re-tiled by device kernels (az_net_commit_device); host tensors / numpy arrays take the host fold."""
        items = [(k, v) for k, v in state_dict.items() if not k.endswith("num_batches_tracked")]
        if items and all(isinstance(v, torch.Tensor) and v.is_cuda for _, v in items):
            keep = []
            for k, v in items:
                t = v.detach().to(torch.float32).contiguous()
                keep.append(t)  # alive until the copies have been queued on this stream
                check(lib().az_net_set_tensor_device(self.h, k.encode(), t.data_ptr(), t.numel(), _stream_ptr()))
            check(lib().az_net_commit_device(self.h, _stream_ptr()))
            return
        for k, v in items:
            a = v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
            a = np.ascontiguousarray(a, dtype=np.float32)
            check(lib().az_net_set_tensor(self.h, k.encode(), a.ctypes.data, a.size))
        check(lib().az_net_commit(self.h, _stream_ptr()))

    def forward(self, x):
        """x: float32 CUDA tensor [B, H*W] of canonical boards (player*grid). Returns (probs [B,A], v [B])."""
        x = x.contiguous().view(-1, self.H * self.W)
        assert x.is_cuda and x.dtype == torch.float32
        B = x.shape[0]
        probs = torch.empty((B, self.A), dtype=torch.float32, device=x.device)
        v = torch.empty(B, dtype=torch.float32, device=x.device)
        check(lib().az_net_forward(self.h, x.data_ptr(), B, probs.data_ptr(), v.data_ptr(), _stream_ptr()))
        return probs, v

    def forward_dyn(self, x, count, probs=None, v=None):
        """az_net_forward_dyn: evaluates only the first min(count[0], B) rows of x; `count` is an int32 CUDA tensor (the engine's
        leaf counter on the hot path: the launch is sized for B, the rows really present are read on the device).  Rows beyond
        the count are left as they were in `probs` / `v`."""
        x = x.contiguous().view(-1, self.H * self.W)
        assert x.is_cuda and x.dtype == torch.float32 and count.is_cuda and count.dtype == torch.int32
        B = x.shape[0]
        if probs is None:
            probs = torch.empty((B, self.A), dtype=torch.float32, device=x.device)
        if v is None:
            v = torch.empty(B, dtype=torch.float32, device=x.device)
        check(lib().az_net_forward_dyn(self.h, x.data_ptr(), count.data_ptr(), B, probs.data_ptr(), v.data_ptr(), _stream_ptr()))
        return probs, v

    def flops_per_board(self):
        return int(lib().az_net_flops_per_board(self.h))

    def time_stage(self, stage, B, iters=20):
        ms = C.c_float()
        check(lib().az_net_time_stage(self.h, stage, B, iters, _stream_ptr(), C.byref(ms)))
        return float(ms.value)

    def stage_kernel(self, stage, B):
        """name of the kernel stage 0..3 launches for B boards (az_net_stage_kernel)"""
        buf = C.create_string_buffer(64)
        check(lib().az_net_stage_kernel(self.h, stage, B, buf, 64))
        return buf.value.decode()

    def profile(self, enable=True):
        """bracket every stage launch of the following forwards with HIP events (az_net_profile)"""
        check(lib().az_net_profile(self.h, 1 if enable else 0))

    def profile_read(self):
        """-> {kernel: (total ms, launches)} since profile(True)"""
        ms, n = (C.c_double * 8)(), (C.c_int64 * 8)()
        check(lib().az_net_profile_read(self.h, ms, n))
        return {k: (ms[i], n[i]) for i, k in enumerate(PROFILE_SLOTS)}

    def profile_overhead_ms(self):
        """correction to subtract per launch from profile_read totals: 0 since every launch carries its own start / stop events
        (rounds 1-3 recorded events between the launches and calibrated an empty interval)"""
        ms = C.c_double()
        check(lib().az_net_profile_overhead(self.h, C.byref(ms)))
        return float(ms.value)

    def close(self):
        if getattr(self, "h", None):
            lib().az_net_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown: module globals may already be gone
            pass


# az_net_profile_read's slots: one per kernel family (a slot's mean is what rocprofv3 lists for that kernel)
PROFILE_SLOTS = ["k_trunk2", "k_gemm fc1", "k_gemm fc2", "k_heads", "k_trunk", "small fc1", "small fc2", "k_trunk_q"]


# ---------------------------------------------------------------------------------------------- engine
class SelfPlayEngine:
    """n_slots concurrent self-play games in lock-step on one GPU (AlphaZeroTrainer.self_play, trainer.py:215-273)."""

    def __init__(self, game_id, H, W, n_slots, n_sim, net=None, dirichlet_alpha=0.03, dirichlet_epsilon=0.25,
                 temp_max_step=4, temp_min_step=4, tie_mode=TIE_RANDOM, noise_mode=NOISE_PHILOX, evaluator=EVAL_NET,
                 seed=0, node_capacity=None, max_plies=None, sample_capacity=None):
        cells = H * W
        if max_plies is None:
            max_plies = 2 * cells if game_id == 0 else cells + 1
        if node_capacity is None:
            # per pool: the subtree kept at a move + everything one search allocates (the tree is compacted
            # into the slot's other pool at every move); exhaustion is reported, never silent
            node_capacity = max(4096, min(1 << 17, 96 * n_sim))
        if sample_capacity is None:
            sample_capacity = n_slots * max_plies
        self.cfg = EngineCfg(game_id, H, W, n_slots, n_sim,
                             -1.0 if dirichlet_alpha is None else dirichlet_alpha,
                             -1.0 if dirichlet_epsilon is None else dirichlet_epsilon,
                             temp_max_step, temp_min_step, tie_mode, noise_mode, evaluator, seed, node_capacity, max_plies,
                             sample_capacity)
        self.net = net
        self.A = H * W + 1 if game_id == 0 else (W if game_id == 1 else 9)
        self.cells = cells
        h = C.c_void_p()
        check(lib().az_engine_create(C.byref(self.cfg), net.h if net is not None else None, _stream_ptr(), C.byref(h)))
        self.h = h

    def run(self, n_games, first_game_id=0):
        """plays n_games to completion; returns the samples as a dict of CUDA tensors (copies)."""
        check(lib().az_engine_run(self.h, first_game_id, n_games))
        return self.samples()

    def samples(self, copy=True):
        n = C.c_int64()
        ps = [C.c_void_p() for _ in range(5)]
        check(lib().az_engine_samples(self.h, C.byref(n), *[C.byref(p) for p in ps]))
        S = n.value
        H, W = self.cfg.H, self.cfg.W
        out = {"state": _wrap(ps[0].value, (S, H, W), torch.int8, self), "pi": _wrap(ps[1].value, (S, self.A), torch.float32, self),
               "z": _wrap(ps[2].value, (S,), torch.int8, self), "meta": _wrap(ps[3].value, (S, 4), torch.int32, self),
               "visits": _wrap(ps[4].value, (S, self.A), torch.int32, self)}
        return {k: v.clone() for k, v in out.items()} if copy else out

    def stats(self):
        st = EngineStats()
        check(lib().az_engine_get_stats(self.h, C.byref(st)))
        return {f: getattr(st, f) for f, _ in st._fields_}

    # finer-grained control ------------------------------------------------------------------
    def set_roots(self, grids, players, game_ids=None, plies=None):
        g = np.ascontiguousarray(grids, np.int8).reshape(-1, self.cells)
        p = np.ascontiguousarray(players, np.int8)
        gi = np.ascontiguousarray(game_ids, np.uint32) if game_ids is not None else None
        pl = np.ascontiguousarray(plies, np.int32) if plies is not None else None
        check(lib().az_engine_set_roots(self.h, g.ctypes.data, p.ctypes.data, gi.ctypes.data if gi is not None else None,
                                        pl.ctypes.data if pl is not None else None, len(p)))

    def search(self, n_sim):
        check(lib().az_engine_search(self.h, n_sim))

    def search_begin(self, n_sim):
        """queues the search and returns; search_end() waits for it (another engine may search in between: the arena's two players)"""
        check(lib().az_engine_search_begin(self.h, n_sim))

    def pair_with(self, other):
        """puts `other`'s stream on a hardware queue of its own so that overlapped searches of the two engines run side by side"""
        check(lib().az_engine_pair(self.h, other.h))

    def search_end(self):
        check(lib().az_engine_search_end(self.h))

    def advance(self):
        check(lib().az_engine_advance(self.h))

    def play(self, actions):
        """Board.play_move + MCT.change_root with externally chosen moves for slots 0..len(actions)-1"""
        a = np.ascontiguousarray(actions, np.int32)
        st = np.zeros(len(a), np.int32)
        check(lib().az_engine_play(self.h, a.ctypes.data, len(a), st.ctypes.data))

    # arena support -------------------------------------------------------------------------------
    def set_sides(self, sides):
        s = np.ascontiguousarray(sides, np.int8)
        check(lib().az_engine_set_sides(self.h, s.ctypes.data, len(s)))

    def best_moves(self):
        a = np.zeros(self.cfg.n_slots, np.int32)
        check(lib().az_engine_best_moves(self.h, a.ctypes.data))
        return a

    def baseline_moves(self, kind, seed=0):
        a = np.zeros(self.cfg.n_slots, np.int32)
        check(lib().az_engine_baseline_moves(self.h, {"random": 0, "greedy": 1}[kind], seed, a.ctypes.data))
        return a

    def root_status(self):
        G = self.cfg.n_slots
        pl = np.zeros(G, np.int8); ov = np.zeros(G, np.uint8); wi = np.zeros(G, np.int8); sc = np.zeros(G, np.int32)
        check(lib().az_engine_root_status(self.h, pl.ctypes.data, ov.ctypes.data, wi.ctypes.data, sc.ctypes.data))
        return pl, ov.astype(bool), wi, sc

    def root_children(self, slot):
        a = np.zeros(65, np.int32); n = np.zeros(65, np.int32); q = np.zeros(65, np.float64); p = np.zeros(65, np.float64)
        k, rn = C.c_int32(), C.c_int32()
        check(lib().az_engine_root_children(self.h, slot, a.ctypes.data, n.ctypes.data, q.ctypes.data, p.ctypes.data,
                                            C.byref(k), C.byref(rn)))
        k = k.value
        return a[:k].copy(), n[:k].copy(), q[:k].copy(), p[:k].copy(), rn.value

    def nodes_used(self, slot=0):
        n = C.c_int32()
        check(lib().az_engine_nodes_used(self.h, slot, C.byref(n)))
        return n.value

    def grow_pools(self, node_capacity):
        """re-allocates the tree pools with a larger capacity; the trees are kept"""
        check(lib().az_engine_grow_pools(self.h, node_capacity))
        self.cfg.node_capacity = node_capacity

    def close(self):
        if getattr(self, "h", None):
            lib().az_engine_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
