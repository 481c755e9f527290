"""Multi-GPU GROUP BY: one process per GPU, partial tables exchanged with ONE all-to-all.

The reference is single-process (README.md:20); this is the MI355X-native addition.  GROUP BY shards
on hash(key): every rank aggregates its own row range into a local table (K7), buckets the
*groups* (not the rows) by hash(key) % world, one all-to-all of counts and one of payload move
the buckets over xGMI (torch.distributed backend "nccl" == RCCL), each rank merges the buckets it
received and emits the groups it owns.  Exchange volume is O(groups), never O(rows).

Two drivers of the same protocol:

* `library_communicator` + `Communicator.exchange(agg)` -- the exchange INSIDE the library (dfx_aggregate_exchange:
  grouped ncclSend/ncclRecv on the library's stream, one host read-back); torch.distributed only carries the 128-byte
  RCCL unique id from rank 0 to the other ranks once per process.  This is what bench.py --gpus N runs.
* `exchange_group_partials` -- the three device steps (partial_build / partial_export / partial_import) with the
  collective done by the HOST plumbing (torch.distributed all_to_all_single; gloo on CPU).  The aggregate object only
  needs those three methods, so the CPU tests drive this function over gloo with an oracle-backed stand-in, and
  bench.py falls back to it if RCCL cannot be bound at run time.
"""
from __future__ import annotations

from typing import Sequence


def library_communicator(world: int, rank: int, dist=None):
    """One RCCL communicator owned by the library for this process: rank 0 draws the unique id, torch.distributed (any
    backend) broadcasts its 128 bytes, every rank calls dfx_comm_init on the library's device."""
    from . import execution as ex
    uid = [None]
    if rank == 0:
        try:
            uid[0] = ex.Communicator.unique_id()
        except Exception as e:  # every rank must still take part in the broadcast (and then fail together)
            uid[0] = ("error", str(e))
    if world > 1:
        if dist is None:
            import torch.distributed as dist  # type: ignore
        dist.broadcast_object_list(uid, src=0)
    if isinstance(uid[0], tuple):
        raise RuntimeError("rank 0 could not create the RCCL unique id: " + uid[0][1])
    return ex.Communicator(uid[0], world, rank)


def exchange_group_partials(agg, world: int, device, dist=None, torch=None) -> dict:
    """Runs the exchange step for `agg` (an AggregateRelation-like object). Returns traffic stats."""
    if torch is None:
        import torch  # type: ignore
    if dist is None:
        import torch.distributed as dist  # type: ignore
    n_words, counts = agg.partial_build(world)
    # the collective runs on `cdev`: the GPU itself (RCCL), or the host when the process group is gloo
    # (CPU tests; shared-GPU dry runs of bench.py) -- the library's entry points always take DEVICE pointers
    is_cuda = getattr(device, "type", str(device)) == "cuda" or str(device).startswith("cuda")
    staged = is_cuda and world > 1 and dist.get_backend() == "gloo"
    cdev = torch.device("cpu") if staged else device
    send_counts = torch.tensor(list(counts), dtype=torch.int64, device=cdev)
    recv_counts = torch.empty_like(send_counts)
    if world > 1:
        dist.all_to_all_single(recv_counts, send_counts)
    else:
        recv_counts.copy_(send_counts)
    rc = [int(x) for x in recv_counts.tolist()]
    send = torch.empty(max(1, n_words * sum(counts)), dtype=torch.int64, device=device)
    agg.partial_export(send.data_ptr(), n_words * sum(counts))  # synchronises the library's stream
    recv = torch.empty(max(1, n_words * sum(rc)), dtype=torch.int64, device=device)
    if world > 1:
        src = send.cpu() if staged else send
        dst = torch.empty(max(1, n_words * sum(rc)), dtype=torch.int64, device=cdev) if staged else recv
        dist.all_to_all_single(dst[: n_words * sum(rc)], src[: n_words * sum(counts)],
                               output_split_sizes=[n_words * c for c in rc],
                               input_split_sizes=[n_words * c for c in counts])
        if staged:
            recv.copy_(dst)
    else:
        recv.copy_(send)
    if is_cuda:
        torch.cuda.synchronize()
    agg.partial_import(recv.data_ptr(), rc)
    return {"n_words": n_words, "sent_groups": int(sum(counts)), "received_groups": int(sum(rc)),
            "sent_bytes": int(8 * n_words * sum(counts))}
