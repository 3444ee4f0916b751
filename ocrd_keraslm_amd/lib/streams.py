"""Batched window generation for stateful training: B streams advanced together.

`Rater.train` trains B independent stateful streams per step; round 2 pulled one Python generator per
stream and step (`windows.file_windows`, the behavioural restatement of rating.py:977-1102) -- 7 us of host time per
stream and step, 21.5 ms at 3072 streams against 26 ms of GPU work.  `StreamBatcher` produces the SAME sequence of
batches (window contents, zero-padded tails, the train=True augmentations and their random numbers, the points where a
stream enters a new file) from array state:

  * every file of every stream is read and NFC-normalised ONCE; its code points go through one look-up table gather
    into one id vector (`corpus`), made and kept where the batches are assembled -- in HBM for the HIP engine, so a
    step moves a few small index vectors instead of three [B, T] arrays over PCIe;
  * per step the streams' states (file, next window, pending augmented copies) advance as numpy vectors; only the
    streams that change file or owe an augmented copy (about one in ten) are touched individually;
  * a batch is assembled from (start, valid length, column to zero, context to zero) per stream by one gather and a
    few masks (`assemble`, numpy on the host or torch on the device -- plumbing, not arithmetic).

Random numbers: the generators draw one uniform number per full training window, stream by stream, in batch order
(rating.py:1062-1077); here the draws of a batch come as ONE vector from the same `numpy.random.Generator`, which
consumes the bit stream identically, so both paths give identical batches for the same seed
(tests/test_streams.py runs them side by side)."""
from __future__ import annotations

import numpy as np

from . import windows


class StreamBatcher(object):
    def __init__(self, stream_files, length, c_i, train=False, rng=None, char_degradation=0.01, context_degradation=0.1,
                 on_unmapped=None, device=None, codepoints=None):
        """stream_files: per stream, the list of open text files it cycles through (at least one each);
        codepoints: optional {id(file): uint32 code points of the normalised text} of files the caller has read
        already (`Rater._split_data`)"""
        self.B = len(stream_files)
        self.T = int(length)
        self.train = bool(train)
        self.rng = rng
        self.char_degradation = float(char_degradation)
        self.context_degradation = float(context_degradation)
        self.device = device
        self.c_i = c_i
        self.on_unmapped = on_unmapped
        T = self.T
        # ---- every distinct file once: code points, size, context.  The ids come later (`prepare`): one gather of the
        # code points through a look-up table, where the batches are assembled (on the device for the HIP engine)
        self._file_of = {}
        off = 0
        f_base, f_size, f_ctx, self._parts = [], [], [], []
        for files in stream_files:
            assert files, "a stream needs at least one file"
            for f in files:
                if id(f) in self._file_of:
                    continue
                if codepoints is not None and id(f) in codepoints:
                    cps = codepoints[id(f)]
                else:
                    f.seek(0)
                    cps = windows.codepoints(windows.read_normalize_file(f)[0])
                size = len(cps)
                self._file_of[id(f)] = len(f_base)
                f_base.append(off)
                f_size.append(size)
                f_ctx.append(windows.clamp_context(windows.context_from_filename(f.name)))
                self._parts.append(cps)
                off += size
        self._total = off
        self._corpus = None
        self.n_ctx = len(f_ctx[0]) if f_ctx else 0
        assert all(len(c) == self.n_ctx for c in f_ctx), "all files must have the same number of context variables"
        self.f_base = np.asarray(f_base, dtype=np.int64)
        self.f_size = np.asarray(f_size, dtype=np.int64)
        self.f_ctx = np.asarray(f_ctx, dtype=np.int32).reshape(len(f_base), self.n_ctx)
        # windows of a file (windows.stateful_windows): full ones end at i = T, 2T, ... < size; then one tail if i + 1 < size
        self.f_full = np.array([len(range(T, s, T)) for s in f_size], dtype=np.int64)
        self.f_tail = (self.f_full * T + 1 < self.f_size)
        # ---- per stream: its files (indices), where it stands
        self.s_files = [np.array([self._file_of[id(f)] for f in files], dtype=np.int64) for files in stream_files]
        self.s_nfiles = np.array([len(x) for x in self.s_files], dtype=np.int64)
        self.s_pos = np.full(self.B, -1, dtype=np.int64)       # index into s_files (-1: before the first file)
        self.cur = np.zeros(self.B, dtype=np.int64)            # current file
        self.win = np.full(self.B, 1 << 60, dtype=np.int64)    # full windows already emitted from it (start: "all of them")
        self.tail_done = np.ones(self.B, dtype=bool)           # (start: "file exhausted", so the first batch opens file 0)
        self.pend_char = np.full(self.B, -1, dtype=np.int64)   # owed copy with this input column zeroed
        self.pend_ctx = np.full(self.B, -1, dtype=np.int64)    # owed copy with this context variable zeroed
        self.last_start = np.zeros(self.B, dtype=np.int64)     # corpus offset of the last full window (for its copies)
        self._corpus_dev = None
        self._ctx_dev = None

    # ------------------------------------------------------------------ code points -> ids
    def _report_unmapped(self, positions):
        """`positions`: corpus offsets of code points outside the mapping, ascending"""
        if self.on_unmapped is None:
            return
        for pos in positions:
            k = int(np.searchsorted(self.f_base, pos, side='right')) - 1
            j = int(pos - self.f_base[k])
            self.on_unmapped(chr(int(self._parts[k][j])), j)

    @property
    def corpus(self):
        """all ids on the host (int32; T + 1 trailing zeros so that a gather one past a file's end stays inside)"""
        if self._corpus is None:
            from concurrent.futures import ThreadPoolExecutor
            import os
            lut = windows.full_lookup_table(self.c_i)
            out = np.zeros(self._total + self.T + 1, dtype=np.int32)

            def gather(k):       # (numpy's take releases the interpreter lock: the files spread over the cores)
                if len(self._parts[k]):
                    np.take(lut, self._parts[k], out=out[self.f_base[k]:self.f_base[k] + self.f_size[k]], mode='clip')
            with ThreadPoolExecutor(max(1, min(16, os.cpu_count() or 1))) as pool:
                list(pool.map(gather, range(len(self._parts))))
            miss = np.nonzero(out[:self._total] < 0)[0]
            if len(miss):
                self._report_unmapped(miss)
                out[miss] = 0
            self._corpus = out
        return self._corpus

    def prepare(self):
        """map the code points to ids now (else on the first batch)"""
        if self.device is None:
            self.corpus
        elif self._corpus_dev is None:
            import torch
            dev = self.device
            import warnings
            cps = torch.zeros(self._total + self.T + 1, dtype=torch.int32, device=dev)
            with warnings.catch_warnings():      # (the vectors are views of immutable bytes; they are only read)
                warnings.simplefilter("ignore")
                for k, part in enumerate(self._parts):
                    if len(part):
                        cps[int(self.f_base[k]):int(self.f_base[k]) + len(part)].copy_(
                            torch.from_numpy(part.view(np.int32)), non_blocking=True)
            lut = torch.from_numpy(windows.full_lookup_table(self.c_i)).to(dev)
            ids = lut.index_select(0, cps)
            ids[self._total:] = 0
            miss = torch.nonzero(ids < 0).flatten()
            if miss.numel():
                self._report_unmapped(miss.cpu().numpy())
                ids.clamp_(min=0)
            self._corpus_dev = ids
            self._ar = torch.arange(self.T, dtype=torch.int64, device=dev)

    # ------------------------------------------------------------------ state machine
    def _open_next_file(self, rows, new_file_rows):
        """streams `rows` have used up their file: enter the next one (cyclic); a file without any window is skipped"""
        rows = list(rows)
        guard = 0
        while rows:
            again = []
            for s in rows:
                self.s_pos[s] = (self.s_pos[s] + 1) % self.s_nfiles[s]
                f = self.s_files[s][self.s_pos[s]]
                self.cur[s] = f
                self.win[s] = 0
                self.tail_done[s] = not self.f_tail[f]
                new_file_rows.append(int(s))
                if self.f_full[f] == 0 and not self.f_tail[f]:
                    again.append(s)
            rows = again
            guard += 1
            assert guard <= int(self.s_nfiles.max()) + 1, "a stream has no file with a window in it"

    def next_plan(self):
        """advance every stream by one item; returns (start [B], vlen [B], zero_col [B], zero_ctx [B], ctx [B, C],
        new_file_rows) -- what `assemble` needs, and the streams that entered a new file while this batch was made
        (the reset points of stateful training, callbacks.py:50-60)"""
        B, T = self.B, self.T
        start = np.empty(B, dtype=np.int64)
        vlen = np.full(B, T, dtype=np.int64)
        zero_col = np.full(B, -1, dtype=np.int64)
        zero_ctx = np.full(B, -1, dtype=np.int64)
        new_file_rows = []
        # 1) owed copies of the last full window (the character copy comes first, rating.py:1066-1077)
        has_char = self.pend_char >= 0
        has_ctx = (~has_char) & (self.pend_ctx >= 0)
        start[has_char | has_ctx] = self.last_start[has_char | has_ctx]
        zero_col[has_char] = self.pend_char[has_char]
        self.pend_char[has_char] = -1
        zero_ctx[has_ctx] = self.pend_ctx[has_ctx]
        self.pend_ctx[has_ctx] = -1
        fresh = ~(has_char | has_ctx)
        # 2) streams whose file is used up enter the next one
        done = fresh & (self.win >= self.f_full[self.cur]) & self.tail_done
        if done.any():
            self._open_next_file(np.nonzero(done)[0], new_file_rows)
        # 3) a full window, or the tail
        full = fresh & (self.win < self.f_full[self.cur])
        tail = fresh & ~full
        base = self.f_base[self.cur]
        start[full] = base[full] + self.win[full] * T
        self.last_start[full] = start[full]
        self.win[full] += 1
        start[tail] = base[tail] + self.f_full[self.cur[tail]] * T
        vlen[tail] = self.f_size[self.cur[tail]] - 1 - self.f_full[self.cur[tail]] * T
        self.tail_done[tail] = True
        # 4) the augmentations a full training window may owe (one random number per window, re-scaled and re-used)
        if self.train and full.any():
            k = int(full.sum())
            rand = self.rng.uniform(0, 1, k) if self.rng is not None else np.random.uniform(0, 1, k)
            rows = np.nonzero(full)[0]
            rmax = self.char_degradation
            if rmax > 0:
                hit = (rand > 0) & (rand < rmax)
                self.pend_char[rows[hit]] = ((T - 1) * rand[hit] / rmax).astype(np.int64)
                rand = (rand - rmax) / (1 - rmax)
            rmax = self.context_degradation
            if rmax > 0 and self.n_ctx:
                hit = (rand > 0) & (rand < rmax)
                self.pend_ctx[rows[hit]] = np.minimum((self.n_ctx * rand[hit] / rmax).astype(np.int64), self.n_ctx - 1)
        ctx = self.f_ctx[self.cur]
        return start, vlen, zero_col, zero_ctx, ctx, new_file_rows

    # ------------------------------------------------------------------ assembly
    def assemble_host(self, plan):
        """(x [B, T], ctx [B, T, C], y [B, T]) int32 numpy arrays"""
        start, vlen, zero_col, zero_ctx, ctx, _ = plan
        T = self.T
        ar = np.arange(T, dtype=np.int64)
        pos = start[:, None] + ar[None, :]
        valid = ar[None, :] < vlen[:, None]
        x = np.where(valid, self.corpus[pos], 0).astype(np.int32)
        y = np.where(valid, self.corpus[pos + 1], -1).astype(np.int32)
        rows = np.nonzero(zero_col >= 0)[0]
        x[rows, zero_col[rows]] = 0
        z = np.where(valid[:, :, None], ctx[:, None, :], 0).astype(np.int32)
        rows = np.nonzero(zero_ctx >= 0)[0]
        z[rows, :, zero_ctx[rows]] = 0
        return x, z, y

    def assemble_device(self, plan):
        """the same three arrays as int32 torch tensors on `self.device`, gathered from the corpus in HBM"""
        import torch
        start, vlen, zero_col, zero_ctx, ctx, _ = plan
        dev = self.device
        if self._corpus_dev is None:
            self.prepare()
        T, B = self.T, self.B
        # one small transfer per step: [start | vlen | zero_col | zero_ctx | ctx...]
        pack = np.concatenate([start[:, None], vlen[:, None], zero_col[:, None], zero_ctx[:, None], ctx.astype(np.int64)], axis=1)
        p = torch.from_numpy(np.ascontiguousarray(pack)).to(dev, non_blocking=True)
        pos = p[:, 0:1] + self._ar[None, :]
        valid = self._ar[None, :] < p[:, 1:2]
        x = torch.where(valid, self._corpus_dev[pos], 0)
        y = torch.where(valid, self._corpus_dev[pos + 1], -1)
        zc = p[:, 2]
        x = torch.where((self._ar[None, :] == zc[:, None]), 0, x)      # (zc = -1 matches no column)
        z = torch.where(valid[:, :, None], p[:, None, 4:].to(torch.int32), 0)
        if self.n_ctx:
            car = torch.arange(self.n_ctx, dtype=torch.int64, device=dev)
            z = torch.where((car[None, None, :] == p[:, 3][:, None, None]), 0, z)
        return x.to(torch.int32).contiguous(), z.to(torch.int32).contiguous(), y.to(torch.int32).contiguous()

    def next_batch(self):
        """-> ((x, ctx, y), rows that entered a new file)"""
        plan = self.next_plan()
        return (self.assemble_device(plan) if self.device is not None else self.assemble_host(plan)), plan[5]
