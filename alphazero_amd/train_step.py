"""Host side of the hand-written training step (csrc/az_train.hip): forward in train mode, loss, backward and the
momentum-SGD update of OthelloNet / Connect4Net on device-resident samples, fifteen HIP kernels per step replayed as a graph.

Replaces the batch loop of AlphaZeroTrainer.optimize_network (trainer.py:346-366) and torch.optim.SGD(lr, momentum 0.9,
weight_decay 1e-4) (trainer.py:326); the stock PyTorch loop stays in trainer.py as the checker (sgd_backend = "torch").
torch only owns the memory: every tensor crosses the C ABI as a raw device pointer under its state-dict name.
"""
import ctypes as C

import torch

from ._lib import check, lib


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def supports(module, batch_size):
    """the hand-written step covers the conv nets at batch sizes that are multiples of 16 up to 512 and the TicTacToe MLP at any batch
    size from 2 to 256 (the reference's defaults: 64)"""
    if not hasattr(module, "hip_shape"):
        return False
    gid, H, W = module.hip_shape()
    if gid == 2:
        # TicTacToeNet (tictactoe.py:272-287): 9 -> 9 -> 9 -> (9, 1), 316 parameters
        return 2 <= batch_size <= 256 and sum(p.numel() for p in module.parameters()) == 316
    if not hasattr(module, "conv1"):
        return False
    if gid == 0 and H not in (6, 8):
        return False
    if gid == 1 and not (5 <= H <= 8 and 5 <= W <= 8):
        return False
    # az_trainer_create builds the reference's architecture: 32 channels, fc widths 1024 / 512 (OthelloNet, othello.py:329-333) or
    # 64 / 32 (Connect4Net, connect4.py:357-361) behind 32 * (h - 4) * (w - 4) inputs; any other module trains on the stock step
    ch, cw = getattr(module, "plane", (H, W))
    want = (32, 32 * (ch - 4) * (cw - 4)) + ((1024, 512) if gid == 0 else (64, 32))
    have = (module.conv1.out_channels, module.fc1.in_features, module.fc1.out_features, module.fc2.out_features)
    if have != want or any(getattr(module, f"conv{i}").out_channels != 32 for i in (2, 3, 4)):
        return False
    return batch_size % 16 == 0 and 16 <= batch_size <= 512


class HipTrainStep:
    def __init__(self, module, max_batch):
        gid, H, W = module.hip_shape()
        self.h = C.c_void_p()
        check(lib().az_trainer_create(gid, H, W, max_batch, C.byref(self.h)))
        self.max_batch = max_batch
        self.steps_done = 0
        self._keep = []

    # ------------------------------------------------------------------ parameters in / out (state-dict names, torch layouts)
    def load(self, module):
        for k, v in module.state_dict().items():
            if v.dtype != torch.float32:
                continue  # num_batches_tracked: counted on the host
            t = v.detach().to("cuda", torch.float32).contiguous()
            self._keep.append(t)
            check(lib().az_trainer_load(self.h, k.encode(), t.data_ptr(), t.numel(), _stream()))
        torch.cuda.current_stream().synchronize()  # the staging copies may go now
        self._keep.clear()

    def store(self, module):
        """writes the trained parameters and BatchNorm statistics back into `module` (on whatever device it lives)"""
        sd = module.state_dict()
        for k, v in sd.items():
            if v.dtype != torch.float32:
                v += self.steps_done  # BatchNorm.num_batches_tracked
                continue
            t = v if v.is_cuda and v.is_contiguous() else torch.empty(v.shape, dtype=torch.float32, device="cuda")
            check(lib().az_trainer_store(self.h, k.encode(), t.data_ptr(), t.numel(), _stream()))
            if t is not v:
                torch.cuda.current_stream().synchronize()
                v.copy_(t)
        torch.cuda.current_stream().synchronize()

    # ------------------------------------------------------------------ optimisation
    def begin(self, lr, momentum=0.9, weight_decay=1e-4, dropout=0.3, seed=0):
        check(lib().az_trainer_begin(self.h, lr, momentum, weight_decay, dropout, seed & 0xFFFFFFFF, _stream()))
        self.steps_done = 0

    def set_lr(self, lr):
        check(lib().az_trainer_set_lr(self.h, lr, _stream()))

    def steps(self, state, pi, z, perm, n_steps, batch_size, loss_pi, loss_v):
        """n_steps steps; step s trains on rows perm[s*B:(s+1)*B] of the device-resident samples (state int8 [S, H, W], pi float32
        [S, A], z int8 [S]); losses land in loss_pi[s], loss_v[s] (float32 CUDA tensors).  Asynchronous on the current stream."""
        assert state.is_cuda and state.dtype == torch.int8 and state.is_contiguous()
        assert pi.is_cuda and pi.dtype == torch.float32 and pi.is_contiguous()
        assert z.is_cuda and z.dtype == torch.int8 and perm.is_cuda and perm.dtype == torch.int64 and perm.is_contiguous()
        assert perm.numel() >= n_steps * batch_size and loss_pi.numel() >= n_steps and loss_v.numel() >= n_steps
        assert loss_pi.dtype == torch.float32 and loss_v.dtype == torch.float32 and loss_pi.is_cuda and loss_v.is_cuda
        S = state.shape[0]
        if pi.shape[0] != S or z.shape[0] != S:
            raise ValueError(f"sample arrays disagree on the number of rows: state {S}, pi {pi.shape[0]}, z {z.shape[0]}")
        # the row count crosses the ABI: a permutation entry outside [0, S) is clamped by the kernels and reported by check()
        check(lib().az_trainer_steps(self.h, state.data_ptr(), pi.data_ptr(), z.data_ptr(), S, perm.data_ptr(), n_steps, batch_size,
                                     loss_pi.data_ptr(), loss_v.data_ptr(), _stream()))
        self.steps_done += n_steps

    def check(self):
        """waits for the enqueued steps; ValueError if one of their permutation entries lay outside the sample arrays"""
        check(lib().az_trainer_check(self.h))

    def debug(self, name, shape=None):
        """a workspace buffer of the last step as a CUDA tensor copy (tests)"""
        p, n = C.c_void_p(), C.c_int64()
        check(lib().az_trainer_debug(self.h, name.encode(), C.byref(p), C.byref(n)))
        from .engine import _wrap
        torch.cuda.synchronize()
        t = _wrap(p.value, (n.value,), torch.float32, self).clone()
        return t.view(shape) if shape is not None else t

    def close(self):
        if getattr(self, "h", None):
            lib().az_trainer_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # interpreter shutdown
            pass
