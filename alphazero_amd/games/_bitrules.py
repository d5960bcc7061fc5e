"""Host-side board rules on Python-int bitboards (cell (r,c) <-> bit r*8+c), the same shift-and-mask
formulation as the device code in csrc/az_device.h.  Used by the Board classes (single positions on the host);
the batched engine never calls into this module."""
M64 = (1 << 64) - 1
COL0 = 0x0101010101010101
COL7 = 0x8080808080808080
# (shift, left?, mask) for E, S, SE, SW, W, N, NW, NE
_DIRS = ((1, True, ~COL0 & M64), (8, True, M64), (9, True, ~COL0 & M64), (7, True, ~COL7 & M64),
         (1, False, ~COL7 & M64), (8, False, M64), (9, False, ~COL7 & M64), (7, False, ~COL0 & M64))


def _sh(x, d):
    amt, left, mask = d
    return ((x << amt) & M64 if left else x >> amt) & mask


def pack(grid):
    p1 = m1 = 0
    H, W = grid.shape
    for r in range(H):
        row = grid[r]
        for c in range(W):
            v = row[c]
            if v > 0:
                p1 |= 1 << (r * 8 + c)
            elif v < 0:
                m1 |= 1 << (r * 8 + c)
    return p1, m1


def valid_mask(H, W):
    v = 0
    for r in range(H):
        v |= ((1 << W) - 1) << (r * 8)
    return v


def othello_legal(own, opp, valid):
    empty = ~(own | opp) & valid
    legal = 0
    for d in _DIRS:
        x = _sh(own, d) & opp
        for _ in range(5):
            x |= _sh(x, d) & opp
        legal |= _sh(x, d) & empty
    return legal


def othello_flips(own, opp, mv):
    flips = 0
    for d in _DIRS:
        x = _sh(mv, d) & opp
        for _ in range(5):
            x |= _sh(x, d) & opp
        if _sh(x, d) & own:
            flips |= x
    return flips


def bits(x):
    while x:
        low = x & -x
        yield low.bit_length() - 1
        x ^= low
