"""One transform split over the ranks of a torch.distributed job, one GPU per rank.

The reference runs a transform of any length in one loop nest (ntt.rs:122-176, :178-236;
ntt_goldilock.rs:81-180).  A vector too large or too slow for one GPU is split the way the
reference's own index structure allows: the first f butterfly stages only couple elements whose
indices agree modulo n/2^f, and after them the vector consists of 2^f independent contiguous
blocks (stage u works inside blocks of n/2^(u-1) elements, ntt.rs:129-147).  So a forward
transform is: f local stages on a column slab, ONE all-to-all, the remaining stages on a
contiguous slab; the backward transform is the mirror image.  Layouts (include/nttx.h):

  column layout of rank d: [rows = 2^f][cols] array; row r holds elements
                           r*(n/2^f) + d*cols ... + cols of the vector
  block layout of rank c:  elements [c*n/world, (c+1)*n/world) of the vector

Local stages are `nttx_shard_phase_device` launches on this rank's GPU.  The exchange runs on
device buffers over RCCL (backend "nccl"): with few rows per peer (the N = 2^28 / 8-GPU case has
two) as one group of point-to-point sends and receives, one message per row, each landing where
the next phase reads it; with many rows per peer as one `all_to_all_single` of whole pieces
followed by a reordering copy.  With another backend and GPU buffers (gloo, used by tests where
several ranks share one GPU) both forms run on host copies of the buffers: the same messages, the
same placement, only the transport differs.
"""
import numpy as np

from ._lib import lib
from .dft import _check


def split_geometry(n, world):
    """(local_elems, rows, cols, rows_per_peer) for a transform of n elements over `world` shards:
    the rule nttx_shard_geometry applies, restated so that layouts can be prepared without a plan.
    rows = 2^f with f the stage count of the first pass (log2 n mod 8, or 8 when that is 0)."""
    logn = n.bit_length() - 1
    if n != 1 << logn or logn % 2 or logn < 10:
        raise ValueError("n must be an even power of two >= 2^10")
    if world < 1 or world & (world - 1) or world > 64:
        raise ValueError("number of shards must be a power of two in [1, 64]")
    k = world.bit_length() - 1
    f = logn % 8 or 8
    if f < k or logn < f + k + 4 or logn < k + 12:
        raise ValueError("n is too small for this number of shards")
    return n >> k, 1 << f, n >> (f + k), (1 << f) >> k


def column_slab(x, rank, world, rows):
    """Rank `rank`'s column-layout input, as a flat copy, of the whole vector x (numpy)."""
    n = x.size
    cols = n // (rows * world)
    return np.ascontiguousarray(x.reshape(rows, world, cols)[:, rank, :]).reshape(-1)


def place_column_slab(x, slab, rank, world, rows):
    """Inverse of column_slab: writes rank `rank`'s column-layout slab into the whole vector x."""
    n = x.size
    cols = n // (rows * world)
    x.reshape(rows, world, cols)[:, rank, :] = slab.reshape(rows, cols)


class _DeviceEngine:
    """Local stages on this rank's GPU through the C ABI."""

    def __init__(self, table):
        self.table = table

    def geometry(self, world):
        return self.table.shard_geometry(world)

    def phase(self, buf, direction, phase, shard, world):
        import torch

        if not buf.is_cuda or not buf.is_contiguous() or buf.element_size() != 8:
            raise TypeError("expected a contiguous 64-bit integer tensor on a GPU")
        stream = torch.cuda.current_stream(buf.device).cuda_stream
        _check(lib.nttx_shard_phase_device(self.table._plan, buf.device.index, buf.data_ptr(), direction, phase,
                                           shard, world, stream))


class ShardedTransform:
    """forward(): column layout in, block layout out (bit-reversed order, as forward_inplace).
    backward(): block layout in, column layout out (natural order, scaled by n^-1)."""

    def __init__(self, table, group=None, engine=None):
        import torch.distributed as dist

        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.engine = engine if engine is not None else _DeviceEngine(table)
        self.local, self.rows, self.cols, self.rows_per_peer = self.engine.geometry(self.world)
        self._device_collective = dist.get_backend(group) == "nccl"

    # Elements per collective call.  RCCL 2.26 (torch 2.10 + ROCm 7) silently leaves the tail of a
    # 2 GiB all_to_all_single unwritten (seen with 2^28 int64 on one rank; 2^27 is intact), so larger
    # slabs go in several calls.
    MAX_ELEMS_PER_CALL = 1 << 27

    def _all_to_all(self, recv, send):
        if self.world == 1:
            recv.copy_(send)
            return
        if not (self._device_collective or not send.is_cuda):
            # several ranks on one GPU (tests): the pieces travel through host memory
            h_send, h_recv = send.cpu(), recv.cpu()
            self.dist.all_to_all_single(h_recv, h_send, group=self.group)
            recv.copy_(h_recv)
            return
        if send.numel() <= self.MAX_ELEMS_PER_CALL:
            self.dist.all_to_all_single(recv, send, group=self.group)
            return
        g = self.world
        piece = send.numel() // g                      # elements per peer
        step = max(1, self.MAX_ELEMS_PER_CALL // g)
        s2, r2 = send.view(g, piece), recv.view(g, piece)
        for p0 in range(0, piece, step):               # the same slice of every peer's piece per call
            part = s2[:, p0:p0 + step].contiguous()
            got = part.new_empty(part.shape)
            self.dist.all_to_all_single(got.view(-1), part.view(-1), group=self.group)
            r2[:, p0:p0 + step] = got

    # Up to this many rows per peer the exchange is posted as rows_per_peer grouped sends and
    # receives per peer that land where the next phase reads them; above it (many small messages)
    # one all_to_all_single moves whole pieces and a transposing copy reorders them.
    MAX_ROWS_DIRECT = 16

    def _direct(self, x):
        """Whether the grouped point-to-point exchange applies."""
        return self.world > 1 and self.rows_per_peer <= self.MAX_ROWS_DIRECT

    def _exchange_rows(self, dst, src, to_blocks):
        """The exchange without a reordering pass.  to_blocks: row c*R + rho of my column layout
        (src) -> rank c, which stores what rank d sent at row rho*g + d of its block layout (dst).
        Otherwise the inverse: row rho*g + d of my block layout (src) -> rank d, which stores what
        rank c sent at row c*R + rho of its column layout (dst).  Messages between two ranks are
        posted in the same (rho) order on both sides."""
        g, R, W = self.world, self.rows_per_peer, self.cols
        me = self.rank
        if src.is_cuda and not self._device_collective:
            # several ranks on one GPU (tests): the same messages between host copies of the buffers
            h_dst = dst.cpu()
            self._exchange_rows(h_dst, src.cpu(), to_blocks)
            dst.copy_(h_dst)
            return
        if to_blocks:
            s3, d3 = src.view(g, R, W), dst.view(R, g, W)
            send_of = lambda peer, rho: s3[peer, rho]
            recv_of = lambda peer, rho: d3[rho, peer]
        else:
            s3, d3 = src.view(R, g, W), dst.view(g, R, W)
            send_of = lambda peer, rho: s3[rho, peer]
            recv_of = lambda peer, rho: d3[peer, rho]
        for rho in range(R):
            recv_of(me, rho).copy_(send_of(me, rho))
        ops = []
        for peer in range(g):
            if peer == me:
                continue
            gp = peer if self.group is None else self.dist.get_global_rank(self.group, peer)
            for rho in range(R):
                ops.append(self.dist.P2POp(self.dist.isend, send_of(peer, rho), gp, group=self.group))
                ops.append(self.dist.P2POp(self.dist.irecv, recv_of(peer, rho), gp, group=self.group))
        for req in self.dist.batch_isend_irecv(ops):
            req.wait()

    def _check_buf(self, x):
        if x.numel() != self.local:
            raise ValueError("expected this rank's %d elements, got %d" % (self.local, x.numel()))

    def forward(self, col):
        import torch

        self._check_buf(col)
        g, R, W = self.world, self.rows_per_peer, self.cols
        self.engine.phase(col, 0, 0, self.rank, g)
        recv = torch.empty_like(col)
        if self._direct(col):
            self._exchange_rows(recv, col, True)   # rows land in block-layout order
            blk = recv
        else:
            self._all_to_all(recv, col)            # piece c = my rows [c*R, (c+1)*R) -> rank c
            # received [source d][R][W]; the block layout orders it [R][d][W]
            blk = recv.view(g, R, W).permute(1, 0, 2).contiguous().view(-1) if (g > 1 and R > 1) else recv
        self.engine.phase(blk, 0, 1, self.rank, g)
        return blk

    def backward(self, blk):
        import torch

        self._check_buf(blk)
        g, R, W = self.world, self.rows_per_peer, self.cols
        self.engine.phase(blk, 1, 1, self.rank, g)
        if self._direct(blk):
            col = torch.empty_like(blk)
            self._exchange_rows(col, blk, False)   # rows land in column-layout order
        else:
            send = blk.view(R, g, W).permute(1, 0, 2).contiguous().view(-1) if (g > 1 and R > 1) else blk
            col = torch.empty_like(send)
            self._all_to_all(col, send)            # piece d of mine -> rank d, lands as its rows [c*R, (c+1)*R)
        self.engine.phase(col, 1, 0, self.rank, g)
        return col
