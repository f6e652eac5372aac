"""Multi-rank host logic of particle migration (one process per GPU, torch.distributed).

The data path shards by owning element: rank r owns a contiguous block of elements,
holds the full mesh (reference `Input::FULL` buffering) and keeps the particles whose element it
owns.  After the search, particles whose new element is owned elsewhere move:

    pp_set_unsafe_procs -> pp_ps_migrate_count -> pp_ps_migrate_pack_records
    -> ONE all-to-all-v of packed records (RCCL over xGMI; gloo on CPU in the tests)
    -> pp_ps_rebuild_records (received particles are the rebuild's new particles)

which is SellCSigma::migrate (scs/SCS_migrate.h:5-222) with the T+1 host-staged MPI messages per
peer replaced by one device-resident exchange.  This module contains only the collective part; it
works on any torch tensors (CUDA with backend nccl, CPU with gloo) so it is testable without GPUs.
"""
import numpy as np
import torch
import torch.distributed as dist


def element_block_owners(nelems, world):
    """owner rank of every element: contiguous blocks (numpy int32)"""
    return (np.arange(nelems, dtype=np.int64) * world // max(nelems, 1)).astype(np.int32)


def exchange_counts(send_counts, device, group=None):
    """all-to-all of one int per peer (PS_Comm_Ialltoall, SCS_migrate.h:48)"""
    sc = torch.as_tensor(np.asarray(send_counts, dtype=np.int64), device=device)
    rc = torch.empty_like(sc)
    dist.all_to_all_single(rc, sc, group=group)
    return [int(v) for v in rc.tolist()]


def exchange_records(send, send_counts, group=None):
    """send: uint8 tensor [total_send, record_bytes], rank-major.  Returns (recv, recv_counts)."""
    send_counts = [int(v) for v in send_counts]
    assert send.dim() == 2 and send.shape[0] == sum(send_counts)
    recv_counts = exchange_counts(send_counts, send.device, group)
    recv = torch.empty((sum(recv_counts), send.shape[1]), dtype=send.dtype, device=send.device)
    dist.all_to_all_single(recv, send.contiguous(), output_split_sizes=recv_counts,
                           input_split_sizes=send_counts, group=group)
    return recv, recv_counts


def allreduce_sum(t, group=None):
    """gyroSync's reduceCommArray(SUM) (pumipic_comm.cpp:234-246) on a device tensor"""
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def migrate(capi, ps, new_elems, new_procs, rank, world, group=None, commit=False, scatter=None):
    """GPU path: move particles routed to other ranks and rebuild.  new_elems / new_procs are
    capi.DevArray (capacity ints) as produced by capi.set_unsafe_procs.
    commit: fold updatePtclPositions (x <- x_tgt, x_tgt <- 0) into the records and the rebuild
    instead of a separate pass before the call; scatter = (mesh, maps, outs): the step's gyroScatter
    calls ride behind the rebuild (pp_ps_rebuild_scatter)."""
    mesh, maps, outs = scatter if scatter is not None else (None, (), ())
    fused = commit or scatter is not None
    if world == 1:  # SCS_migrate.h:20-25
        if fused:
            capi.rebuild_scatter(ps, mesh, new_elems, list(maps), list(outs), commit=commit)
        else:
            ps.rebuild(new_elems)
        return 0, 0
    counts = capi.migrate_count(ps, new_elems, new_procs, rank, world)
    recb = capi.migrate_record_bytes(ps)
    dev = torch.device("cuda", torch.cuda.current_device())
    send = torch.empty((int(counts.sum()), recb), dtype=torch.uint8, device=dev)
    if commit:
        capi.migrate_pack_records_commit(ps, new_elems, new_procs, rank, world, counts, send.data_ptr())
    else:
        capi.migrate_pack_records(ps, new_elems, new_procs, rank, world, counts, send.data_ptr())
    capi.sync()                      # library stream -> visible to the collective's stream
    recv, recv_counts = exchange_records(send, counts, group)
    torch.cuda.synchronize()
    if fused:
        capi.rebuild_records_scatter(ps, new_elems, int(recv.shape[0]), recv.data_ptr(), mesh, list(maps),
                                     list(outs), commit=commit)
    else:
        capi.rebuild_records(ps, new_elems, int(recv.shape[0]), recv.data_ptr())
    return int(counts.sum()), int(recv.shape[0])
