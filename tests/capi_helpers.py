"""Thin wrappers that call the C-ABI with torch device tensors as the memory plumbing."""
import numpy as np
import torch

from semantic_search_kd_amd import _native


def stream():
    return int(torch.cuda.current_stream().cuda_stream)


def tile_corpus(lib, corpus: np.ndarray, normalize: bool = False) -> torch.Tensor:
    n = corpus.shape[0]
    tiled = torch.empty(max(int(lib.sskd_index_tiled_bytes(n)) // 4, 1), dtype=torch.float32, device="cuda")
    rows = torch.from_numpy(np.ascontiguousarray(corpus, np.float32)).cuda()
    _native.check(lib.sskd_index_add_rows(rows.data_ptr(), n, int(normalize), tiled.data_ptr(), 0, stream()))
    return tiled


def capi_search(lib, tiled: torch.Tensor, n_rows: int, queries: np.ndarray, k: int, id_offset: int = 0):
    nq = queries.shape[0]
    q = torch.from_numpy(np.ascontiguousarray(queries, np.float32)).cuda()
    out_s = torch.full((nq, k), float("nan"), dtype=torch.float32, device="cuda")
    out_i = torch.full((nq, k), -7, dtype=torch.int64, device="cuda")
    ws_bytes = int(lib.sskd_index_search_workspace_bytes(n_rows, max(nq, 1), k))
    ws = torch.empty(max(ws_bytes, 1), dtype=torch.uint8, device="cuda")
    _native.check(
        lib.sskd_index_search(
            tiled.data_ptr(), n_rows, q.data_ptr(), nq, k, id_offset,
            out_s.data_ptr(), out_i.data_ptr(), ws.data_ptr(), ws.numel(), stream(),
        )
    )
    torch.cuda.synchronize()
    return out_s.cpu().numpy(), out_i.cpu().numpy()
