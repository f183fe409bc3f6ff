"""Multi-GPU plumbing: one process per GPU; the read-only flat graph is loaded once by rank 0 and replicated with one
`torch.distributed.broadcast` (RCCL over xGMI) per flat buffer; long reads are sharded by batch ticket with no collective
on the data path. This is the MI355X counterpart of the reference's only distribution scheme -- a replicated index and
chunked long reads (reference: Ratatosk_nf/Ratatosk.nf:277-299; threads + tickets inside a node, src/Ratatosk.cpp:727-906).
"""
import ctypes as C

from . import api


def load_graph_replicated(fasta_gz, rtsk, k, rank, world, device, lib_path=None, report=None, n_threads=None):
    """Returns an api.Graph whose HBM image is valid on `device` of every rank. report (a dict, optional) receives on rank 0 what the
    replication cost: parse + flatten seconds and threads, bytes and seconds of every broadcast."""
    import os
    import time
    if n_threads is None:  # rank 0 parses and flattens alone while the others wait: it may use the host's threads
        n_threads = max(1, min(64, os.cpu_count() or 1))
    if world <= 1:
        return api.Graph(fasta_gz, rtsk, k, device=device, lib_path=lib_path, n_threads=n_threads)
    import torch
    import torch.distributed as dist
    L = api.load_library(lib_path)
    sim = lib_path is not None
    if not sim:
        torch.zeros(1, device=torch.device("cuda", device))  # torch's HIP context before the library's first HIP call on rank 0 (the other order has failed to find the GPU)
    n_buf = L.rtk_graph_n_buffers(None)
    sizes = (C.c_uint64 * n_buf)()
    info = api.RtkGraphInfo()
    g = api.Graph.__new__(api.Graph)
    g.L, g.k, g.h = L, k, C.c_void_p()
    if rank == 0:
        t0 = time.time()
        # (the simulator builds its tables on the host; on a GPU rank 0 builds them in its own HBM at upload time, and their sizes are known after that)
        g._check(L.rtk_graph_load2(api._b(fasta_gz), api._b(rtsk), k, n_threads, 0 if sim else api.RTK_LOAD_DEVICE_TABLES, C.byref(g.h)))
        if not sim:
            g._check(L.rtk_graph_upload(g.h, device))
        t_load = time.time() - t0
        g._check(L.rtk_graph_buffer_bytes(g.h, sizes, n_buf))
        g._check(L.rtk_graph_get_info(g.h, C.byref(info)))
        meta = [[int(sizes[i]) for i in range(n_buf)], bytes(info)]
    else:
        g._check(L.rtk_graph_shell(k, C.byref(g.h)))
        meta = [None, None]
    dist.broadcast_object_list(meta, src=0)
    for i in range(n_buf):
        sizes[i] = meta[0][i]
    C.memmove(C.byref(info), meta[1], C.sizeof(info))
    dev = torch.device("cpu") if sim else torch.device("cuda", device)
    L.rtk_graph_attach_buffers.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64), C.c_int, C.POINTER(api.RtkGraphInfo)]
    if rank == 0 and not sim:  # resident image -> the tensors the broadcasts read, one buffer at a time (a whole-genome graph does not fit twice)
        tensors = []
        for i in range(n_buf):
            tensors.append(torch.empty(max(8, int(sizes[i])), dtype=torch.uint8, device=dev))
            g._check(L.rtk_graph_move_buffer(g.h, i, C.c_void_p(tensors[i].data_ptr()), max(8, int(sizes[i]))))
    else:
        tensors = [torch.empty(max(8, int(sizes[i])), dtype=torch.uint8, device=dev) for i in range(n_buf)]
        ptrs = (C.c_void_p * n_buf)(*[t.data_ptr() for t in tensors])
        g._check(L.rtk_graph_attach_buffers(g.h, device, ptrs, sizes, n_buf, C.byref(info)))
        if rank == 0:
            g._check(L.rtk_graph_upload(g.h, device))  # host image -> the attached buffers
    secs = []
    for t in tensors:  # one large contiguous broadcast per buffer; ring/tree over xGMI is per-link bound
        t0 = time.time()
        dist.broadcast(t, src=0)
        if not sim:
            torch.cuda.synchronize()
        secs.append(time.time() - t0)
    if report is not None and rank == 0:
        tot_b, tot_s = sum(int(sizes[i]) for i in range(n_buf)), sum(secs)
        report.update({"ranks": world, "load_flatten_s": round(t_load, 3), "load_threads": n_threads, "bytes_per_buffer": [int(sizes[i]) for i in range(n_buf)],
                       "broadcast_s_per_buffer": [round(x, 5) for x in secs], "broadcast_total_s": round(tot_s, 4), "broadcast_GBps": round(tot_b / tot_s / 1e9, 2) if tot_s > 0 else None})
    if rank != 0:
        g._check(L.rtk_graph_adopt_device(g.h))
    g._tensors = tensors  # keep the HBM buffers alive as long as the graph
    return g


def shard_tickets(n_tickets, rank, world):
    """Batch tickets owned by `rank` (round-robin, like the reference's ticket dispenser: src/Ratatosk.cpp:755)."""
    return [i for i in range(n_tickets) if i % world == rank]
