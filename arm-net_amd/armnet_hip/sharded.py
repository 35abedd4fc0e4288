"""Row-sharded embedding lookup over RCCL (no reference counterpart — SURVEY.md §8e).

Rank r of R owns the table rows ``{i : i % R == r}`` (local index ``i // R``; the modulo partition
balances skewed ids).  One lookup of a rank's ``ids [B, F]``:

    route      armnet_shard_route_ids: counting sort of the B*F ids by owner          (HIP, this rank)
    exchange   all_gather of the R counts  ->  every rank knows the R x R split matrix (tiny)
    exchange   all_to_all_single of int32 local row indices                             (RCCL over xGMI)
    gather     armnet_gather_scale_f32(vals=NULL): owner reads its rows                 (HIP, HBM-bound)
    exchange   all_to_all_single of the rows (E*4 bytes per DISTINCT id with de-dup)    (RCCL over xGMI)
    consume    armnet_fused_fwd_f32 with table = received rows, ids = perm (int32)      (no un-permute pass)

The arithmetic of the fused block is untouched: the sharded result is bit-equal to the single-GPU one.
`ops` abstracts the two device kernels so that the routing logic can be exercised by world_size-2
gloo tests on CPU with a test double (tests/test_sharded_gloo.py); the product default is HipShardOps.
"""
import torch
import torch.distributed as dist

from . import native


class HipShardOps:
    """Device kernels of the sharded lookup (C ABI).  CPU tensors are rejected by the binding."""

    def __init__(self):
        self._ws = None          # workspace of the de-duplicating route, reused across steps

    def route(self, ids_flat, R, nfeat, dedup=False, id_status=None):
        """-> counts [R], send_local [>= sum(counts)], perm [n].  With dedup every distinct id is sent once.
        id_status (int32[1], optional) is OR-ed with 1 when an id lies outside [0, nfeat) (such ids read row 0)."""
        n = ids_flat.numel()
        dev = ids_flat.device
        counts = torch.empty(R, device=dev, dtype=torch.int32)
        send_local = torch.empty(n, device=dev, dtype=torch.int32)
        perm = torch.empty(n, device=dev, dtype=torch.int32)
        if dedup:
            need = native.shard_route_unique_ws_bytes(R, nfeat)
            if self._ws is None or self._ws.numel() < need or self._ws.device != dev:
                self._ws = torch.empty(need, device=dev, dtype=torch.uint8)
            native.shard_route_unique_ids(n, ids_flat, R, nfeat, counts, send_local, perm, self._ws, id_status)
        else:
            ws = torch.empty(max(native.shard_route_ws_bytes(n, R), 4), device=dev, dtype=torch.uint8)
            native.shard_route_ids(n, ids_flat, R, nfeat, counts, send_local, perm, ws, id_status)
        return counts, send_local, perm

    def gather(self, local_idx, table_local):
        out = torch.empty(local_idx.numel(), table_local.shape[1], device=table_local.device, dtype=torch.float32)
        if local_idx.numel():
            native.gather_scale(local_idx.numel(), table_local.shape[1], local_idx, None, table_local, out)
        return out


def shard_rows(full_table, rank, world):
    """Local shard of a full [nfeat, E] table under the modulo partition."""
    return full_table[rank::world].contiguous()


class RowShardedTable:
    """One rank's shard of the embedding table + the lookup protocol above."""

    def __init__(self, table_local, nfeat, group=None, ops=None, dedup="auto"):
        self.dedup = dedup        # True / False / "auto" (de-duplicate when the batch is >= 1/8 of the table)
        self.micro_batches = 1    # > 1: sharded_arm_block overlaps the exchange of slice m+1 with the kernel of slice m
        self.table_local = table_local
        self.nfeat = int(nfeat)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.ops = ops if ops is not None else HipShardOps()
        # gloo has no device all-to-all: with that backend (debugging, the 2-processes-on-1-GPU test) the two exchanges
        # are staged through host memory; RCCL ("nccl") exchanges device buffers directly
        self._via_host = bool(dist.is_initialized() and dist.get_backend(group) == "gloo" and table_local.is_cuda)
        expect = (self.nfeat - self.rank + self.world - 1) // self.world
        if table_local.shape[0] != expect:
            raise ValueError(f"rank {self.rank}: shard has {table_local.shape[0]} rows, expected {expect}")

    def lookup(self, ids, id_status=None):
        """ids [B, F] (this rank's samples) -> (rows [B*F, E] in send order, perm [B*F] int32).
        id_status: optional int32[1] device flag, set when an id is out of range (the caller raises IndexError)."""
        R = self.world
        flat = ids.reshape(-1).contiguous()
        n = flat.numel()
        dedup = (8 * n >= self.nfeat) if self.dedup == "auto" else bool(self.dedup)
        if n == 0:                                     # an empty slice still takes part in the exchanges
            counts = torch.zeros(R, device=flat.device, dtype=torch.int32)
            send_local = torch.empty(0, device=flat.device, dtype=torch.int32)
            perm = torch.empty(0, device=flat.device, dtype=torch.int32)
        else:
            counts, send_local, perm = (self.ops.route(flat, R, self.nfeat, dedup=dedup, id_status=id_status)
                                        if id_status is not None else self.ops.route(flat, R, self.nfeat, dedup=dedup))
        E = self.table_local.shape[1]
        if R == 1 and not dist.is_initialized():
            n_send = int(counts.sum().item()) if dedup else n
            return self.ops.gather(send_local[:n_send], self.table_local), perm
        # split matrix: row q = what rank q sends to each owner
        if self._via_host:
            allc = torch.empty(R * R, dtype=torch.int32)
            dist.all_gather_into_tensor(allc, counts.cpu(), group=self.group)
        else:
            allc = torch.empty(R * R, device=counts.device, dtype=torch.int32)
            dist.all_gather_into_tensor(allc, counts, group=self.group)
        m = allc.view(R, R).cpu()                      # the one host sync of the step
        send_counts = m[self.rank].tolist()
        recv_counts = m[:, self.rank].tolist()
        recv_idx = torch.empty(sum(recv_counts), device=flat.device, dtype=torch.int32)
        n_send = sum(send_counts)                      # == n without de-duplication
        self._all_to_all(recv_idx, send_local[:n_send], recv_counts, send_counts)
        rows_out = self.ops.gather(recv_idx, self.table_local)
        rows_in = torch.empty(n_send, E, device=flat.device, dtype=torch.float32)
        self._all_to_all(rows_in, rows_out, send_counts, recv_counts)
        return rows_in, perm

    def _all_to_all(self, out, inp, out_splits, in_splits):
        if self._via_host:
            h = torch.empty(out.shape, dtype=out.dtype)
            dist.all_to_all_single(h, inp.cpu(), out_splits, in_splits, group=self.group)
            out.copy_(h)
        else:
            dist.all_to_all_single(out, inp, out_splits, in_splits, group=self.group)


def sharded_arm_block(shard, ids, vals, q_fold, values, bn_scale, bn_shift, alpha, n_iter=50,
                      write_clamped_vals=True, flags=0, micro_batches=None, check_ids=False):
    """Fused a2..a9 with the table row-sharded over the process group.  Returns out [B, O, E].

    With micro_batches > 1 the batch is processed in slices whose lookups (routing, the two exchanges, the owner-side
    gather) run on a side stream, so the exchange of slice m+1 overlaps the fused kernel of slice m.  Every rank must
    use the same number of slices (the collectives pair up slice by slice).  Default: `shard.micro_batches` (1).
    With check_ids an out-of-range id raises IndexError like the replicated path (one host sync at the end of the call;
    the routing kernels flag it, the lookup itself reads row 0 for such an id)."""
    from .block import arm_block_forward
    B, F = vals.shape
    M = int(micro_batches if micro_batches is not None else getattr(shard, "micro_batches", 1))
    status = torch.zeros(1, device=ids.device, dtype=torch.int32) if (check_ids and ids.is_cuda) else None

    def finish(out):
        if status is not None and int(status.item()) != 0:
            raise IndexError("index out of range in self")
        return out

    if M <= 1 or not vals.is_cuda:
        rows, perm = shard.lookup(ids, status)
        return finish(arm_block_forward(perm.view(B, F), vals, rows, q_fold, values, bn_scale, bn_shift, alpha,
                                        n_iter=n_iter, write_clamped_vals=write_clamped_vals, check_ids=False,
                                        flags=flags))
    O, E = q_fold.shape
    out = torch.empty(B, O, E, device=vals.device, dtype=torch.float32)
    compute = torch.cuda.current_stream()
    side = getattr(shard, "_side_stream", None)
    if side is None:
        side = shard._side_stream = torch.cuda.Stream()
    side.wait_stream(compute)                       # ids / vals were produced on the compute stream
    step = (B + M - 1) // M
    for m in range(M):
        lo, hi = m * step, min(B, (m + 1) * step)   # every rank runs M slices, possibly an empty last one
        with torch.cuda.stream(side):
            rows, perm = shard.lookup(ids[lo:hi], status)
            ready = side.record_event()
        if hi > lo:
            compute.wait_event(ready)
            rows.record_stream(compute)
            perm.record_stream(compute)
            arm_block_forward(perm.view(hi - lo, F), vals[lo:hi], rows, q_fold, values, bn_scale, bn_shift, alpha,
                              n_iter=n_iter, write_clamped_vals=write_clamped_vals, check_ids=False, flags=flags,
                              out=out[lo:hi])
    return finish(out)
