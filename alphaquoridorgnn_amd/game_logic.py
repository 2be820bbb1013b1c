"""Quoridor game logic -- drop-in for the reference's game_logic.py, backed by the HIP library.

`State` keeps the reference's constructor, public mutable attributes and method names
(game_logic.py:15-54, :96-117, :195-196, :359-395).  The cheap bookkeeping (next, is_done, to_array ...) is
host Python like the reference; everything that walks the board -- legal_actions(), legal_actions_pos(),
legal_actions_wall() -- runs the gfx950 kernel (aqg_legal_actions) on a batch of one.  Batched callers
should use `legal_actions_batch` / `next_batch` / `status_batch` on device tensors of state72 records.
"""
import numpy as np
import torch

from . import _lib
from .constants import BOARD_SIZE, NUM_WALLS, NUM_PLIES_FOR_DRAW

MOVEMENT_DIRECTIONS = [(-1, 0), (1, 0), (0, -1), (0, 1)]  # U, D, L, R (game_logic.py:11)
STATE72 = 72
MAX_LEGAL = _lib.MAX_LEGAL


def num_actions(board_size=BOARD_SIZE):
    return board_size ** 2 + 2 * (board_size - 1) ** 2


# ------------------------------------------------------------------ batched device API
def legal_actions_batch(states72, board_size=BOARD_SIZE, want_mask=True):
    """states72: uint8 [B,72] device tensor -> (mask u8 [B,A] or None, order u8 [B,136], count i32 [B])."""
    dev = _lib.require_gpu(states72.device)
    lib = _lib.load()
    B = states72.shape[0]
    A = num_actions(board_size)
    mask = torch.empty((B, A), dtype=torch.uint8, device=dev) if want_mask else None
    order = torch.empty((B, MAX_LEGAL), dtype=torch.uint8, device=dev)
    count = torch.empty((B,), dtype=torch.int32, device=dev)
    _lib.check(lib.aqg_legal_actions(board_size, _lib.ptr(states72), B, _lib.ptr(mask), _lib.ptr(order), _lib.ptr(count),
                                     _lib.stream_ptr(dev)), "aqg_legal_actions")
    return mask, order, count


def next_batch(states72, actions, board_size=BOARD_SIZE):
    dev = _lib.require_gpu(states72.device)
    out = torch.empty_like(states72)
    actions = actions.to(device=dev, dtype=torch.int32).contiguous()
    _lib.check(_lib.load().aqg_state_next(board_size, _lib.ptr(states72), _lib.ptr(actions), states72.shape[0],
                                          _lib.ptr(out), _lib.stream_ptr(dev)), "aqg_state_next")
    return out


def status_batch(states72, board_size=BOARD_SIZE, plies_for_draw=NUM_PLIES_FOR_DRAW):
    """flags u8 [B]: bit0 = is_lose, bit1 = is_draw."""
    dev = _lib.require_gpu(states72.device)
    out = torch.empty((states72.shape[0],), dtype=torch.uint8, device=dev)
    _lib.check(_lib.load().aqg_state_status(board_size, _lib.ptr(states72), states72.shape[0], plies_for_draw,
                                            _lib.ptr(out), _lib.stream_ptr(dev)), "aqg_state_status")
    return out


def pack_state72(player, enemy, walls, plies_played, board_size):
    r = np.zeros(STATE72, dtype=np.uint8)
    r[0], r[1] = int(player[0]), int(player[1])
    r[2], r[3] = int(enemy[0]), int(enemy[1])
    w = np.asarray(walls, dtype=np.uint8)
    r[4:4 + len(w)] = w
    r[68] = plies_played & 0xFF
    r[69] = (plies_played >> 8) & 0xFF
    r[70] = board_size
    return r


# ------------------------------------------------------------------ reference-shaped single state
class State:
    """
    :param walls: 1 by (N-1)^2 int list from the current player's perspective; 0 none, 1 horizontal, 2 vertical,
        slot i = 2x2 block with top-left tile (i // (N-1), i % (N-1)).
    :param player: [position, walls left], position in the player's own frame.
    :param enemy: [position, walls left], position in the ENEMY's own frame.
    :param plies_played: plies played so far.
    """

    def __init__(self, board_size=BOARD_SIZE, num_walls=NUM_WALLS, player=None, enemy=None, walls=None, plies_played=0):
        self.N = board_size
        N = self.N
        if N % 2 == 0:
            raise ValueError('The board size must be an odd number.')
        self.player = player if player is not None else [0] * 2
        self.enemy = enemy if enemy is not None else [0] * 2
        self.walls = walls if walls is not None else [0] * ((N - 1) ** 2)
        self.plies_played = plies_played
        if player is None or enemy is None:
            init_pos = N * (N - 1) + N // 2
            self.player[0] = init_pos
            self.player[1] = num_walls
            self.enemy[0] = init_pos
            self.enemy[1] = num_walls

    # -- cheap host-side bookkeeping (game_logic.py:43-54, :96-100, :359-395)
    def is_lose(self):
        return self.enemy[0] // self.N == 0

    def is_draw(self):
        from . import constants
        draw = NUM_PLIES_FOR_DRAW if self.N == BOARD_SIZE else constants.board_params(self.N)[1]
        return self.plies_played >= draw

    def is_done(self):
        return self.is_lose() or self.is_draw()

    def to_array(self):
        return [list(self.player), list(self.enemy), list(self.walls)]

    def is_first_player(self):
        return self.plies_played % 2 == 0

    def rotate_walls(self):
        self.walls = list(self.walls)[::-1]

    @classmethod
    def from_record(cls, rec):
        """State of a state72 record (the layout in include/aqgnn.h)."""
        N = int(rec[70])
        nw = (N - 1) ** 2
        return cls(board_size=N, player=[int(rec[0]), int(rec[1])], enemy=[int(rec[2]), int(rec[3])],
                   walls=[int(x) for x in rec[4:4 + nw]], plies_played=int(rec[68]) | (int(rec[69]) << 8))

    def next(self, action):
        """game_logic.py:366-391 -- move the pawn or set the wall, turn the board by 180 degrees, swap the players -- by the rule
        header the kernels compile, in its host instantiation (aqg_host_next): one transition code for device and host."""
        import ctypes
        rec, out = self.record(), np.empty(STATE72, dtype=np.uint8)
        if _lib.load().aqg_host_next(self.N, rec.ctypes.data_as(ctypes.c_void_p), int(action), out.ctypes.data_as(ctypes.c_void_p)) != 0:
            raise ValueError(f"action {action} is not an action of a {self.N}x{self.N} board")
        return State.from_record(out)

    # -- board-walking queries: HIP kernel, batch of one
    def record(self):
        return pack_state72(self.player, self.enemy, self.walls, self.plies_played, self.N)

    def _device_record(self, rec=None):
        dev = _lib.require_gpu()
        return torch.from_numpy(self.record() if rec is None else rec).to(dev).unsqueeze(0)

    def legal_actions(self):
        """Ordered like the reference: pawn moves (U,D,L,R / jumps), then per wall slot H, V (game_logic.py:103-117)."""
        _, order, count = legal_actions_batch(self._device_record(), self.N, want_mask=False)
        n = int(count.item())
        return [int(a) for a in order[0, :n].cpu().numpy()]

    def legal_actions_pos(self, pos):
        """Pawn destinations from `pos` (game_logic.py:120-192): the pawn-move prefix of the legal list of the
        same position with the mover placed on `pos`."""
        rec = self.record()
        rec[0] = int(pos)
        rec[1] = 0          # no walls left -> the list is the pawn moves only
        _, order, count = legal_actions_batch(self._device_record(rec), self.N, want_mask=False)
        n = int(count.item())
        return [int(a) for a in order[0, :n].cpu().numpy()]

    def legal_actions_wall(self, pos):
        """Legal wall placements at slot `pos` (game_logic.py:195-357)."""
        rec = self.record()
        if rec[1] == 0:
            rec[1] = 1      # legality of a slot does not depend on the wall count (:113 gates the loop only)
        mask, _, _ = legal_actions_batch(self._device_record(rec), self.N, want_mask=True)
        N = self.N
        m = mask[0].cpu().numpy()
        out = []
        if m[N * N + pos]:
            out.append(N * N + pos)
        if m[N * N + (N - 1) ** 2 + pos]:
            out.append(N * N + (N - 1) ** 2 + pos)
        return out

    def __str__(self):
        """Compact ASCII render (the reference's :398-456 debug print is out of scope; this is our own)."""
        N = self.N
        me, en = self.player[0], N * N - 1 - self.enemy[0]
        if not self.is_first_player():
            pass
        rows = []
        for x in range(N):
            rows.append(" ".join("P" if x * N + y == me else ("E" if x * N + y == en else ".") for y in range(N)))
        walls = [f"{'H' if w == 1 else 'V'}{i}" for i, w in enumerate(self.walls) if w]
        return "\n".join(rows) + f"\nwalls: {' '.join(walls) or '-'}  left: {self.player[1]}/{self.enemy[1]}  ply {self.plies_played}"
