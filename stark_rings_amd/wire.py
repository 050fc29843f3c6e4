"""ark-serialize framing of ring-element containers around the device codec.

The per-element bytes come from the C ABI (`sr_serialize_batch` / `sr_deserialize_batch`, include/stark_rings_hip.h); this
module adds what the reference's containers add around them:

  Vec<R>            u64 little-endian length, then the elements            (ark-serialize Vec<T>)
  Matrix<R>         its Vec<Vec<R>>                                        crates/linear_algebra/src/matrix.rs:111-145
  SparseMatrix<R>   nrows u64, ncols u64, then Vec<Vec<(R, usize)>>        crates/linear_algebra/src/sparse_matrix.rs:158-200
                    (a pair is its members in order; usize travels as u64)

Ring elements are numpy uint64 words in the reference's memory image (rings.py); wire data is numpy uint8.  Errors the
reference reports as SerializationError (short input, a coefficient >= p) raise RingError.
"""
import numpy as np

from .rings import RingError


def _u64(v):
    return np.frombuffer(int(v).to_bytes(8, "little"), dtype=np.uint8)


def _read_u64(wire, pos):
    if pos + 8 > wire.size:
        raise RingError("deserialize: unexpected end of input")
    return int.from_bytes(wire[pos:pos + 8].tobytes(), "little"), pos + 8


def elem_bytes(ring):
    return ring.degree * ring.wire_coeff_bytes


def serialize_vec(ring, elems):
    """Vec<R> for a flat batch of ring elements."""
    batch = ring._batch_of(elems.size)
    return np.concatenate([_u64(batch), ring.serialize(elems)])


def deserialize_vec(ring, wire, pos=0):
    """-> (elements, position after the vector)"""
    wire = np.ascontiguousarray(wire, dtype=np.uint8)
    n, pos = _read_u64(wire, pos)
    end = pos + n * elem_bytes(ring)
    if end > wire.size:
        raise RingError("deserialize: unexpected end of input")
    return ring.deserialize(wire[pos:end]), end


def serialize_matrix(ring, m, nrows, ncols):
    """Matrix<R> (row-major flat batch of nrows * ncols elements): outer length, then per row its length and elements."""
    w, eb = ring.words_per_elem, elem_bytes(ring)
    if m.size != nrows * ncols * w:
        raise RingError("serialize: DifferentLengths")
    payload = ring.serialize(m).reshape(nrows, ncols * eb) if nrows else np.zeros((0, 0), dtype=np.uint8)
    out = np.empty(8 + nrows * (8 + ncols * eb), dtype=np.uint8)
    out[:8] = _u64(nrows)
    body = out[8:].reshape(nrows, 8 + ncols * eb)
    if nrows:
        body[:, :8] = _u64(ncols)
        body[:, 8:] = payload
    return out


def deserialize_matrix(ring, wire):
    """-> (flat elements, nrows, ncols); ncols is the first row's length as in matrix.rs:139-143, and a ragged matrix (which
    the reference would build without complaint and fail on later) is refused."""
    wire = np.ascontiguousarray(wire, dtype=np.uint8)
    nrows, pos = _read_u64(wire, 0)
    eb = elem_bytes(ring)
    chunks, ncols = [], 0
    for r in range(nrows):
        n, pos = _read_u64(wire, pos)
        if r == 0:
            ncols = n
        elif n != ncols:
            raise RingError("deserialize: ragged matrix rows")
        if pos + n * eb > wire.size:
            raise RingError("deserialize: unexpected end of input")
        chunks.append(wire[pos:pos + n * eb])
        pos += n * eb
    flat = np.concatenate(chunks) if chunks else np.zeros(0, dtype=np.uint8)
    return ring.deserialize(flat), nrows, ncols


def serialize_sparse(ring, nrows, ncols, rows):
    """SparseMatrix<R>; rows as CyclotomicRing.spmv_ntt takes them: one list per row of (element words, column)."""
    w, eb = ring.words_per_elem, elem_bytes(ring)
    nnz = sum(len(r) for r in rows)
    vals = np.zeros(max(nnz * w, 1), dtype=np.uint64)[:nnz * w]
    j = 0
    for row in rows:
        for elem, _ in row:
            vals[j * w:(j + 1) * w] = elem
            j += 1
    payload = ring.serialize(vals) if nnz else np.zeros(0, dtype=np.uint8)
    out = np.empty(24 + 8 * len(rows) + nnz * (eb + 8), dtype=np.uint8)
    out[0:8], out[8:16], out[16:24] = _u64(nrows), _u64(ncols), _u64(len(rows))
    pos, j = 24, 0
    for row in rows:
        out[pos:pos + 8] = _u64(len(row))
        pos += 8
        for _, col in row:
            out[pos:pos + eb] = payload[j * eb:(j + 1) * eb]
            out[pos + eb:pos + eb + 8] = _u64(col)
            pos += eb + 8
            j += 1
    return out


def deserialize_sparse(ring, wire):
    """-> (nrows, ncols, rows) with rows as serialize_sparse takes them"""
    wire = np.ascontiguousarray(wire, dtype=np.uint8)
    w, eb = ring.words_per_elem, elem_bytes(ring)
    nrows, pos = _read_u64(wire, 0)
    ncols, pos = _read_u64(wire, pos)
    nlists, pos = _read_u64(wire, pos)
    chunks, shape = [], []
    for _ in range(nlists):
        n, pos = _read_u64(wire, pos)
        cols = []
        for _ in range(n):
            if pos + eb + 8 > wire.size:
                raise RingError("deserialize: unexpected end of input")
            chunks.append(wire[pos:pos + eb])
            col, pos = _read_u64(wire, pos + eb)
            cols.append(col)
        shape.append(cols)
    vals = ring.deserialize(np.concatenate(chunks)) if chunks else np.zeros(0, dtype=np.uint64)
    rows, j = [], 0
    for cols in shape:
        row = []
        for col in cols:
            row.append((vals[j * w:(j + 1) * w], col))
            j += 1
        rows.append(row)
    return nrows, ncols, rows


def sparse_layout(ring, row_ptr):
    """For a CSR matrix held on the device: (total wire bytes, byte offset of every stored element, positions and values of
    the u64 framing words apart from the column indices, byte offset of every column word) -- what serialize_sparse_dev needs."""
    eb = elem_bytes(ring)
    row_ptr = np.asarray(row_ptr, dtype=np.int64)
    nlists = row_ptr.size - 1
    counts = np.diff(row_ptr)
    row_start = 24 + 8 * np.arange(nlists, dtype=np.int64) + row_ptr[:-1] * (eb + 8)   # position of each row's length word
    within = np.arange(int(row_ptr[-1]), dtype=np.int64) - np.repeat(row_ptr[:-1], counts)
    elem_off = np.repeat(row_start + 8, counts) + within * (eb + 8)
    total = 24 + 8 * nlists + int(row_ptr[-1]) * (eb + 8)
    return total, elem_off, row_start, counts, elem_off + eb


def serialize_sparse_dev(ring, nrows, ncols, vals, cols, row_ptr, stream=None):
    """SparseMatrix<R> from device-resident CSR (vals: int64 CUDA tensor of nnz elements, cols int32, row_ptr int64, the
    layout of CyclotomicRing.spmv_ntt_dev): the elements are encoded straight into their framed positions on the device
    (`d_offsets` of sr_serialize_batch_dev); only the length and column words are written around them.  Returns a uint8 CUDA
    tensor holding exactly the bytes serialize_sparse produces."""
    import torch

    if elem_bytes(ring) % 8:
        raise RingError("serialize_sparse_dev: element size is not a multiple of 8 bytes (BabyBear, D = 1); use serialize_sparse")
    total, elem_off, row_start, counts, col_off = sparse_layout(ring, row_ptr.cpu().numpy())
    dev = vals.device
    wire = torch.empty(total, dtype=torch.uint8, device=dev)
    words = wire.view(torch.int64)            # every framing word sits at a multiple of 8
    head = torch.tensor([nrows, ncols, counts.size], dtype=torch.int64, device=dev)
    words[:3] = head
    if counts.size:
        words[torch.from_numpy(row_start // 8).to(dev)] = torch.from_numpy(counts.astype(np.int64)).to(dev)
    if elem_off.size:
        words[torch.from_numpy(col_off // 8).to(dev)] = cols.to(torch.int64)
        ring.serialize_dev(wire, vals, offsets=torch.from_numpy(elem_off).to(dev), stream=stream)
    return wire
