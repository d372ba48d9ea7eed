"""Host-side mirror of rnascan's scan layer on top of the HIP engine.

Same names, argument meaning and table layout as the reference functions
(rnascan/rnascan.py): ``scan`` (:258-275), ``scan_all`` (:278-286),
``scan_averaged_structure`` (:293-315), ``scan_main`` (:335-413), ``combine``
(:416-434), ``_add_match_id`` (:329-332) -- but a whole batch of records goes
through ONE kernel launch instead of one Python loop per window.

The engine is the only thing that computes scores.  ``HipEngine`` (the default
and the only engine in this package) calls libpfmscan through ctypes and raises
when the library or the GPU is missing; there is no CPU path here.
"""
import os

import numpy as np
import pandas as pd

from . import _lib, fasta, pack, table
from .pssm import PSSM

SEQ_COLUMNS = ["Sequence_ID", "Description", "Motif_ID", "Start", "End", "Sequence", "LogOdds"]
PIPELINE_CHUNK = int(os.environ.get("RNASCAN_PIPELINE_CHUNK", str(1 << 24)))     # positions per chunk of the host pipeline
LIBRARY_MAX_M = 64                                                                # PFMSCAN_MAX_M: wider PFMs are scanned one by one (the plain kernel)
LETTER_LIBRARY_MAX_M = 32                                                         # structure-letter libraries (k_library8): wider PFMs one by one
PIPELINE_MIN = 2 * PIPELINE_CHUNK                                                 # shorter streams are staged whole


class HipEngine(object):
    """Scores packed streams on one MI355X through the C ABI (include/pfmscan.h).

    The last stream stays staged on the device: scanning the same ``Stream`` object
    with another motif (a multi-PFM library) re-uses it without any upload."""

    MOTIF_CACHE = 8                 # motifs kept on the device between calls (a CLI run scans every batch with the same few)

    def __init__(self, device=0):
        self.ctx = _lib.Context(device)
        self._staged = None
        self._staged2 = None        # the Stream whose codes2 are staged beside the staged codes
        self._library = None        # (key, _lib.Library): the tables of the last library stay on the device across batches
        self._motifs = {}           # (letter table bytes, structure PSSM bytes) -> _lib.Motif, in order of last use

    def close(self):
        if self._library is not None:
            self._library[1].close()
            self._library = None
        for mo in self._motifs.values():
            mo.close()
        self._motifs = {}
        self.ctx.close()

    def _motif(self, letter_table=None, struct_pssm=None):
        """the device-resident operands of a motif, kept across calls: with them stay the threshold-dependent tables the
        hits kernels derive on the host (credit tables and their survivor prediction), which a batch-by-batch CLI run
        would otherwise rebuild for every batch"""
        lt = None if letter_table is None else np.ascontiguousarray(letter_table, dtype=np.float64)
        sp = None if struct_pssm is None else np.ascontiguousarray(struct_pssm, dtype=np.float64)
        key = (None if lt is None else (lt.shape, lt.tobytes()), None if sp is None else (sp.shape, sp.tobytes()))
        mo = self._motifs.pop(key, None)
        if mo is None:
            mo = self.ctx.motif(lt, sp)
            while len(self._motifs) >= self.MOTIF_CACHE:
                self._motifs.pop(next(iter(self._motifs))).close()
        self._motifs[key] = mo                              # most recently used last
        return mo

    def _stage(self, stream):
        if self._staged is None or self._staged[0] is not stream or self._staged[1] != self.ctx.scratch_gen:
            self._staged = None
            self._staged2 = None
            gen = self.ctx.stage(stream.codes, stream.profile)
            self._staged = (stream, gen)

    def scan(self, stream, letter_table=None, struct_pssm=None):
        """all window scores, position aligned -> (float32 seq | None, float64 struct | None)"""
        motif = self._motif(letter_table, struct_pssm)
        self._stage(stream)
        return self.ctx.scan_staged(motif)

    def pwm_calculate(self, sequence, matrix):
        """``_pwm.calculate(sequence, matrix)`` (_pwm.c:79-121): str + float64 [m][4] (A,C,G,U) -> float32 [n]"""
        self._staged = None
        return self.ctx.pwm_calculate(sequence, matrix)

    def scan_letters_f64(self, stream, letter_table):
        """generic-alphabet letter scores in fp64 (matrix.py:25-43)"""
        self._staged = None
        return self.ctx.scan_letters_f64_host(self._motif(letter_table, None), stream.codes)

    def hits_letters_f64(self, stream, letter_table, thr):
        """positions (sorted) whose fp64 letter score exceeds thr (matrix.py:25-43 + rnascan.py:263) -> (pos, score float64)"""
        motif = self._motif(letter_table, None)
        self._stage(stream)
        return self.ctx.hits_letters_f64_staged(motif, thr)

    def hits_pair(self, stream, seq_table, struct_table, thr_seq, thr_struct):
        """positions (sorted) where the letters of ``stream.codes`` score above thr_seq (float32 of the fp64 sum) AND the
        letters of ``stream.codes2`` above thr_struct (fp64) -- the two-FASTA combined scan (rnascan.py:416-434 joins the
        two hit tables) -> (pos, seq float32, struct float64)"""
        mo_seq, mo_st = self._motif(seq_table, None), self._motif(struct_table, None)
        self._stage(stream)
        if self._staged2 is not stream:
            self.ctx.stage_codes2(stream.codes2)
            self._staged2 = stream
        return self.ctx.hits_pair_staged(mo_seq, mo_st, thr_seq, thr_struct)

    def hits(self, stream, letter_table=None, struct_pssm=None, thr_seq=-np.inf, thr_struct=-np.inf, one_shot=True):
        """positions (sorted) whose scores exceed the thresholds -> (pos, seq | None, struct | None).
        A stream that is not on the device yet and is longer than PIPELINE_MIN positions (a memory-mapped profile store,
        a large batch) goes through the chunked pipeline: upload and scan overlap, device scratch stays two chunks --
        when this is the ONLY scan of the stream (``one_shot``).  A caller that loops over motifs says so: the stream is
        then staged once for all of them (the pipeline leaves nothing staged, every motif would upload it again)."""
        motif = self._motif(letter_table, struct_pssm)
        staged = self._staged is not None and self._staged[0] is stream and self._staged[1] == self.ctx.scratch_gen
        if one_shot and not staged and stream.n_pos > PIPELINE_MIN:
            self._staged = None
            return self.ctx.hits_pipeline_host(motif, stream.codes if letter_table is not None else None,
                                               stream.profile if struct_pssm is not None else None, thr_seq, thr_struct,
                                               PIPELINE_CHUNK)
        self._stage(stream)
        return self.ctx.hits_staged(motif, thr_seq, thr_struct)


def _library_hits(self, stream, letter_tables, struct_pssms, thr_seq, thr_struct=None, one_shot=True):
    """hits of EVERY motif of a library in one pass over the stream: letter_tables [n][m][8] or None,
    struct_pssms [n][m][7] or None, thresholds scalar or [n] -> (pos, motif index, seq float32 | None,
    struct float64 | None) sorted by (pos, motif index).  With letter tables: k_library; structure PSSMs alone:
    k_profile_lib (the profile is read once, whatever the library's size)."""
    T = None if letter_tables is None else np.ascontiguousarray(letter_tables, dtype=np.float64)
    P = None if struct_pssms is None else np.ascontiguousarray(struct_pssms, dtype=np.float64)
    key = ((T if T is not None else P).shape, None if T is None else T.tobytes(), None if P is None else P.tobytes())
    if self._library is None or self._library[0] != key:
        if self._library is not None:
            self._library[1].close()
            self._library = None
        self._library = (key, self.ctx.library(T, P))
    staged = self._staged is not None and self._staged[0] is stream and self._staged[1] == self.ctx.scratch_gen
    if one_shot and not staged and stream.n_pos > PIPELINE_MIN:
        # a long stream that only this library will scan (a memory-mapped profile store + the codes of its records):
        # chunked, upload beside scan, two chunks of device scratch
        self._staged = None
        return self.ctx.library_hits_pipeline_host(self._library[1], stream.codes if T is not None else None,
                                                   stream.profile if P is not None else None, thr_seq, thr_struct, PIPELINE_CHUNK)
    self._stage(stream)
    return self.ctx.library_hits_staged(self._library[1], thr_seq, thr_struct)


HipEngine.library_hits = _library_hits


def _library_hits_letters(self, stream, seq_tables, struct_tables, thr_seq, thr_struct):
    """hits of EVERY motif of a LETTER library in one pass (SURVEY 8f N1 x N4): struct_tables [n][m][8] (up to 7 letters,
    fp64 scores) over ``stream.codes`` -- a structure-letter library, k_library8 -- or, with seq_tables [n][m][8]
    (nucleotides), pair k = (sequence PFM k over ``stream.codes``, structure-letter PFM k over ``stream.codes2``): the
    two-FASTA library.  -> (pos, motif index, seq float32 | None, struct float64) sorted by (pos, motif index)."""
    T = None if seq_tables is None else np.ascontiguousarray(seq_tables, dtype=np.float64)
    S = np.ascontiguousarray(struct_tables, dtype=np.float64)
    key = ("letters", S.shape, None if T is None else T.tobytes(), S.tobytes())
    if self._library is None or self._library[0] != key:
        if self._library is not None:
            self._library[1].close()
            self._library = None
        self._library = (key, self.ctx.library(T, struct_letters=S))
    self._stage(stream)
    if T is not None and self._staged2 is not stream:
        self.ctx.stage_codes2(stream.codes2)
        self._staged2 = stream
    return self.ctx.library_hits_staged(self._library[1], thr_seq, thr_struct)


HipEngine.library_hits_letters = _library_hits_letters


def _first_motif(pssm):
    """rnascan.py:262 / :298 -- only the FIRST motif of the dict is scanned."""
    return list(pssm.items())[0]


def _select(engine, stream, m, letter_table, struct_pssm, thr_seq, thr_struct, one_shot=True):
    """hits of a stream; an infinite threshold (-m ' -inf') would make every window
    a hit, so that case takes the all-scores kernel and filters on the host with the
    same strict `>` (NaN and -inf never pass: rnascan.py:263, :310)."""
    if np.isneginf(thr_seq) and np.isneginf(thr_struct):
        sq, st = engine.scan(stream, letter_table, struct_pssm)
        keep = stream.window_mask(m)
        if sq is not None:
            keep &= sq.astype(np.float64) > thr_seq
        if st is not None:
            keep &= st > thr_struct
        pos = np.flatnonzero(keep)
        return pos, (None if sq is None else sq[pos]), (None if st is None else st[pos])
    pos, sq, st = engine.hits(stream, letter_table, struct_pssm, thr_seq, thr_struct, one_shot=one_shot)
    if letter_table is None:                 # no codes -> no separator poisoning: drop windows that
        rec, start = stream.locate(pos)      # run over a record end on the host
        ok = start + m <= stream.lengths[rec]
        pos, st = pos[ok], st[ok]
    return pos, sq, st


def _select_letters_f64(engine, stream, m, letter_table, thr):
    """hits of a generic-alphabet letter scan (Python floats in the reference: fp64, no float32 cast; matrix.py:25-43)
    -> (pos, score).  Finite thresholds are decided on the device; at -inf every window with a finite score is a row,
    so the all-scores kernel runs and the same strict `>` drops NaN and -inf here."""
    if np.isneginf(thr) or not hasattr(engine, "hits_letters_f64"):
        full = engine.scan_letters_f64(stream, letter_table)
        pos = np.flatnonzero(stream.window_mask(m) & (full > thr))
        return pos, full[pos]
    return engine.hits_letters_f64(stream, letter_table, thr)


# ---------------------------------------------------------------------------
# sequence / letter-string scans
# ---------------------------------------------------------------------------
class _RnaBatch(object):
    """A batch of records in stream form: the codes (one separator after each record), and where the
    strings of a record are found when a hit needs them.  From a fasta.FastaSlice the letters are mapped and packed
    natively from the mapped file (pfmscan_fasta_encode); from Records by preprocess_seq (rnascan.py:186-197) +
    pack.encode_rna.  Either way a hit's ``Sequence`` is read back from the codes: a hit window holds the four
    nucleotides only (a foreign letter makes the window NaN), and preprocess_seq upper-cases and turns T into U,
    which is what codes 0..3 -> ``ACGU`` gives.

    ``letters``: a generic alphabet (structure strings, ``EHTBLRM``) instead of the nucleotides.  Its records are NOT
    transcribed or upper-cased (rnascan.py:186-197), the scores ignore the case (matrix.py:31) and the ``Sequence``
    column shows the string as written (rnascan.py:272): the codes keep the case in bit 3 (pack.CASE_BIT), which the
    kernels do not read."""

    def __init__(self, records, letters=None):
        self.letters = letters
        if letters is None:
            packed = records.pack_rna() if hasattr(records, "pack_rna") else None
        else:
            packed = records.pack_letters(pack.letter_lut(letters, keep_case=True)) if hasattr(records, "pack_letters") else None
        self.spans = None                      # (file bytes, id spans, header spans): strings the writer copies itself
        if packed is not None:
            self.codes, self.offsets, self.lengths = packed
            self.ids, self.descriptions = records.ids, records.descriptions
            self.spans = records.span_tables()
        else:
            recs = list(records)
            if letters is None:
                coded = [pack.encode_rna(fasta.preprocess_seq(r.seq, True)) for r in recs]
            else:
                coded = [pack.encode_letters(r.seq, letters, keep_case=True) for r in recs]
            st = pack.pack(coded) if recs else None
            self.codes = st.codes if st else np.zeros(0, dtype=np.uint8)
            self.offsets = st.offsets if st else np.zeros(0, dtype=np.int64)
            self.lengths = st.lengths if st else np.zeros(0, dtype=np.int64)
            self.ids = [r.id for r in recs]
            self.descriptions = [r.description for r in recs]

    def __len__(self):
        return len(self.offsets)

    def select(self, keep):
        """the batch restricted to records ``keep`` (ascending indices)"""
        if len(keep) == len(self):
            return self
        out = _RnaBatch([], self.letters)
        parts = [self.codes[int(self.offsets[i]):int(self.offsets[i] + self.lengths[i]) + 1] for i in keep]
        out.codes = np.concatenate(parts) if parts else np.zeros(0, dtype=np.uint8)
        out.lengths = self.lengths[keep]
        out.offsets = np.zeros(len(keep), dtype=np.int64)
        if len(keep) > 1:
            out.offsets[1:] = np.cumsum(out.lengths[:-1] + 1)
        out.ids = [self.ids[i] for i in keep]
        out.descriptions = [self.descriptions[i] for i in keep]
        if self.spans is not None:
            out.spans = (self.spans[0], self.spans[1][keep], self.spans[2][keep])
        return out

    def windows(self, pos, m):
        """the ``Sequence`` column of hits at stream positions ``pos``"""
        if self.letters is None:
            return table.Windows(self.codes, pos, m, pack.RNA_LETTERS)
        return table.Windows(self.codes, pos, m, self.letters, cased=True)

    def id_column(self, rec):
        return table.Spans(self.spans[0], self.spans[1], rec) if self.spans is not None else table.Indexed(self.ids, rec)

    def description_column(self, rec):
        return table.Spans(self.spans[0], self.spans[2], rec) if self.spans is not None else table.Indexed(self.descriptions, rec)


def _finish(tables, order, sort_keys, columns):
    """per-motif hit tables (compact columns, table.py) -> one table.  One table is already in (record, Start, motif)
    order; several are put in record order, then sort_values(sort_keys) inside a record (rnascan.py:286).  Returns
    the compact columns when asked for (the streaming writer formats them natively), else a DataFrame."""
    if not tables:
        return pd.DataFrame(columns=order)
    if len(tables) == 1:
        return tables[0] if columns else table.to_frame(tables[0], order)
    df = pd.concat([table.to_frame(t, ["_rec"] + order) for t in tables], ignore_index=True)
    df = df.sort_values(["_rec"] + sort_keys, kind="stable").reset_index(drop=True)
    return df[order]


def scan_records(engine, records, pssm, letters, minscore, columns=False):
    """Batch form of scan_all (rnascan.py:278-286) + the per-record tagging of
    scan_main (:401-402): all records in one launch.

    records: iterable of fasta.Record (or a fasta.FastaSlice); letters: the alphabet's letters (``GAUC``
    for RNA, ``EHTBLRM`` for structure strings).  Returns the hit table with the
    reference's columns, rows in record order then by Start (``columns=True``: possibly as compact columns,
    see table.py)."""
    is_rna = fasta.is_rna_letters(letters)
    if not hasattr(records, "__len__"):
        records = list(records)                                 # any iterable of Records
    if not len(records):
        return pd.DataFrame(columns=SEQ_COLUMNS)
    # sorted(alphabet.letters) for nucleotides (matrix.py:57); a generic alphabet in its own order
    order = pack.RNA_LETTERS if is_rna else letters
    batch = _RnaBatch(records, None if is_rna else letters)    # packed natively from the mapped file when it can be
    stream = pack.Stream(batch.codes, None, batch.offsets, batch.lengths)       # (and staged on the device once)
    id_column, description_column = batch.id_column, batch.description_column

    def fragments(pos, rec, start, m):
        return batch.windows(pos, m)
    tables = []

    def rows(motif_ids, m, pos, mo, logodds):
        rec, start = stream.locate(pos)
        return {"_rec": rec, "Sequence_ID": id_column(rec), "Description": description_column(rec),
                "Motif_ID": motif_ids[0] if mo is None else table.Indexed(motif_ids, mo),
                "Start": start + 1, "End": start + m, "Sequence": fragments(pos, rec, start, m), "LogOdds": logodds}

    # An RNA library with a finite threshold goes through the one-pass library kernel, one launch per PFM width
    # (the reference's dict only ever holds one motif, rnascan.py:262).  Motifs are ordered by id, so
    # the kernel's (position, motif index) order is the table's (Start, Motif_ID) order.
    by_width = {}
    for motif_id in sorted(pssm.keys()) if len(pssm) > 1 else list(pssm.keys()):
        by_width.setdefault(pssm[motif_id].length, []).append(motif_id)
    for m, mids in by_width.items():
        if is_rna and len(mids) > 1 and np.isfinite(float(minscore)) and m <= LIBRARY_MAX_M:
            T = np.stack([pssm[i].letter_table(order) for i in mids])
            pos, mo, sq, _ = engine.library_hits(stream, T, None, float(minscore), one_shot=len(by_width) == 1)
            tables.append(rows(mids, m, pos, mo, np.round(sq, 3)))
            continue
        if not is_rna and len(mids) > 1 and np.isfinite(float(minscore)) and m <= LETTER_LIBRARY_MAX_M and \
                hasattr(engine, "library_hits_letters"):
            # a structure-letter library (`-q library structs.fa`): one pass per width too (k_library8), scores in fp64
            T = np.stack([pssm[i].letter_table(order) for i in mids])
            pos, mo, _, sc = engine.library_hits_letters(stream, None, T, None, float(minscore))
            tables.append(rows(mids, m, pos, mo, _lib.round_decimals(sc, 3)))
            continue
        for motif_id in mids:
            tab = pssm[motif_id].letter_table(order)
            if is_rna:
                pos, sq, _ = _select(engine, stream, m, tab, None, float(minscore), -np.inf, one_shot=len(pssm) == 1)
                logodds = np.round(sq, 3)                      # round(np.float32, 3) stays float32 (rnascan.py:273)
            else:
                pos, sc = _select_letters_f64(engine, stream, m, tab, float(minscore))
                logodds = _lib.round_decimals(sc, 3)            # round(Python float, 3), rnascan.py:273
            tables.append(rows([motif_id], m, pos, None, logodds))
    return _finish(tables, SEQ_COLUMNS, ["Start", "Motif_ID"], columns)


def scan(engine, pssm, seq, letters, minscore):
    """rnascan.py:258-275 for one (already preprocessed) sequence string:
    list of [motif_id, Start, End, fragment, round(score, 3)]."""
    df = scan_records(engine, [fasta.Record("", "", seq)], pssm, letters, minscore)
    return [[r.Motif_ID, int(r.Start), int(r.End), r.Sequence, r.LogOdds] for r in df.itertuples()]


def scan_all(engine, record, pssm, letters, minscore):
    """rnascan.py:278-286: one record -> [Motif_ID, Start, End, Sequence, LogOdds] sorted by Start."""
    df = scan_records(engine, [record], pssm, letters, minscore)
    return df[["Motif_ID", "Start", "End", "Sequence", "LogOdds"]]


# ---------------------------------------------------------------------------
# averaged-structure scans
# ---------------------------------------------------------------------------
def struct_matrix(pm, file_letters, pairing="aligned"):
    """PSSM operand [m][7] for a profile whose columns are ``file_letters``.

    ``aligned``: column k scores the letter the file says column k holds -- the
    evident intent and what the reference computed where ``DataFrame(dict)``
    sorted its keys (Python 2 / old pandas).  ``positional``: column k is paired
    with the k-th letter the PSSM was filled in (``EHTBLRM``), which is what
    rnascan.py:300-307 does on Python >= 3.6 / pandas >= 0.23."""
    if pairing == "aligned":
        return pm.matrix(file_letters)
    if pairing == "positional":
        return pm.matrix(list(pm.keys())[:len(file_letters)])
    raise ValueError("pairing must be 'aligned' or 'positional'")


FLOAT32_STORAGE_BUDGET = 0.5e-6      # half of north_star's 1e-6 on float scores is spent on storage, the rest stays for the arithmetic


def float32_storage_bound(struct_pssm):
    """Worst-case |score(float32 rows) - score(float64 rows)| of an averaged-structure scan (rnascan.py:302-307) with the
    structure PSSMs ``struct_pssm`` ({id: PSSM}), over EVERY profile whose rows are non-negative and sum to at most 1:
    rounding a row entry p to float32 moves it by at most 2^-24 p, so a row-dot with PSSM row j moves by at most
    2^-24 sum_c p_c |P_jc| <= 2^-24 max_c |P_jc|, and a window by the sum over its rows.  Only FINITE cells count: a
    row-dot that meets a +-inf / NaN cell (p_c > 0 there, or 0 x inf) is replaced by nan_to_num identically in either
    storage -- float32 keeps exact zeros and the sign of every entry above 1.2e-38."""
    worst = 0.0
    for pm in struct_pssm.values():
        P = np.asarray(pm.matrix(list(pm.keys())), dtype=np.float64)
        fin = np.where(np.isfinite(P), np.abs(P), 0.0)
        worst = max(worst, float(fin.max(axis=1).sum()) * 2.0 ** -24)
    return worst


def pick_profile_dtype(requested, struct_pssm):
    """device storage of averaged-structure profile rows: what was asked for, or -- ``auto`` -- float32 when the PFM at
    hand PROVES it within FLOAT32_STORAGE_BUDGET (float32_storage_bound), else float64.  -> (numpy type, bound)"""
    bound = float32_storage_bound(struct_pssm)
    if requested in ("float32", "float64"):
        return np.dtype(requested).type, bound
    return (np.float32 if bound < FLOAT32_STORAGE_BUDGET else np.float64), bound


def _scan_profile_stream(engine, stream, ids, letters, pssm, minscore, pairing, columns=False):
    """hit table of a packed profile stream (no codes): rnascan.py:302-315 for every record.  A library (several
    motifs, pfmutil.py:89-133) with a finite threshold is ONE pass per PFM width over the profile (k_profile_lib),
    instead of the reference's one scan_averaged_structure call per motif; the stream is staged on the device once."""
    tables = []
    thr = float(minscore)

    def rows(motif_ids, m, pos, mo, st):
        rec, start = stream.locate(pos)
        ok = start + m <= stream.lengths[rec]                 # no codes -> no separators: drop windows that run over a record end
        if not ok.all():
            pos, st, rec, start = pos[ok], st[ok], rec[ok], start[ok]
            mo = None if mo is None else mo[ok]
        return {"_rec": rec, "Sequence_ID": table.Indexed(ids, rec), "Description": "",
                "Motif_ID": motif_ids[0] if mo is None else table.Indexed(motif_ids, mo),
                "Start": start + 1, "End": start + m, "Sequence": ".", "LogOdds": st}

    by_width = {}
    for motif_id in sorted(pssm.keys()) if len(pssm) > 1 else list(pssm.keys()):
        by_width.setdefault(pssm[motif_id].length, []).append(motif_id)
    for m, mids in by_width.items():
        if len(mids) > 1 and np.isfinite(thr) and hasattr(engine, "library_hits") and m <= LIBRARY_MAX_M:
            P = np.stack([struct_matrix(pssm[i], list(letters), pairing) for i in mids])
            pos, mo, _, st = engine.library_hits(stream, None, P, None, thr, one_shot=len(by_width) == 1)
            tables.append(rows(mids, m, pos, mo, st))
            continue
        for motif_id in mids:
            P = struct_matrix(pssm[motif_id], list(letters), pairing)
            pos, _, st = _select(engine, stream, m, None, P, -np.inf, thr, one_shot=len(pssm) == 1)      # (drops record-crossing windows itself)
            tables.append(rows([motif_id], m, pos, None, st))
    return _finish(tables, SEQ_COLUMNS, ["Start", "Motif_ID"], columns)


def scan_profiles(engine, named_profiles, pssm, minscore, pairing="aligned", profile_dtype=np.float32, columns=False):
    """Batch form of scan_averaged_structure (rnascan.py:293-315) + the tagging of
    scan_main (:367-374).  named_profiles: list of (Sequence_ID, letters, [L][7])."""
    if not named_profiles:
        return pd.DataFrame(columns=SEQ_COLUMNS)
    letters0 = list(named_profiles[0][1])
    for _, letters, _ in named_profiles:
        if list(letters) != letters0:
            raise ValueError("averaged-structure files disagree on their column order")
    stream = pack.pack(profiles=[p for _, _, p in named_profiles], profile_dtype=profile_dtype)
    return _scan_profile_stream(engine, stream, [n for n, _, _ in named_profiles], letters0, pssm, minscore, pairing, columns)


def scan_store(engine, profile_store, pssm, minscore, pairing="aligned", lo=0, hi=None, columns=False):
    """Same table from a packed profile store (rnascan_amd/store.py): the mapped file
    already IS the stream layout, nothing is parsed or repacked."""
    hi = len(profile_store.ids) if hi is None else hi
    if hi <= lo:
        return pd.DataFrame(columns=SEQ_COLUMNS)
    return _scan_profile_stream(engine, profile_store.stream(lo, hi), profile_store.ids[lo:hi], profile_store.letters,
                                pssm, minscore, pairing, columns)


def scan_averaged_structure(engine, struct_file, pssm, minscore, pairing="aligned", profile_dtype=np.float64):
    """rnascan.py:293-315 for one profile file -> [Motif_ID, Start, End, Sequence, LogOdds]."""
    letters, prof = fasta.read_profile(struct_file)
    df = scan_profiles(engine, [("", letters, prof)], pssm, minscore, pairing, profile_dtype)
    return df[["Motif_ID", "Start", "End", "Sequence", "LogOdds"]]


def scan_profile_dir(engine, directory, pssm, minscore, pairing="aligned", profile_dtype=np.float32):
    """scan_main's directory branch (rnascan.py:348-375, pool form)."""
    fasta.eprint("Scanning averaged secondary structures ")
    files = fasta.list_profiles(directory)
    if len(files) == 0:
        raise IOError("No averaged structure files found")
    parsed = fasta.read_profiles([path for _, path in files])
    named = [(sid, letters, prof) for (sid, _), (letters, prof) in zip(files, parsed)]
    df = scan_profiles(engine, named, pssm, minscore, pairing, profile_dtype)
    fasta.eprint("Processed %d sequences" % len(named))
    return df


# ---------------------------------------------------------------------------
# combined scan: one fused pass instead of two tables + a hash join
# ---------------------------------------------------------------------------
COMBINED_COLUMNS = ["Sequence_ID", "Description.Seq", "Motif_ID.Seq", "Start", "End", "Sequence.Seq", "LogOdds.Seq",
                    "Description.Struct", "Motif_ID.Struct", "Sequence.Struct", "LogOdds.Struct", "LogOdds.SeqStruct"]


def combine(seq_results, struct_results):
    """rnascan.py:416-434: inner join on (Sequence_ID, Start, End); the combined
    score is the sum of the two log-odds (float32 + float64 -> float64)."""
    result = pd.merge(seq_results, struct_results, on=["Sequence_ID", "Start", "End"])
    result = result.rename(columns={"Description_x": "Description.Seq", "Description_y": "Description.Struct",
                                    "Sequence_x": "Sequence.Seq", "Sequence_y": "Sequence.Struct",
                                    "Motif_ID_x": "Motif_ID.Seq", "Motif_ID_y": "Motif_ID.Struct",
                                    "LogOdds_x": "LogOdds.Seq", "LogOdds_y": "LogOdds.Struct"})
    result["LogOdds.SeqStruct"] = result["LogOdds.Seq"] + result["LogOdds.Struct"]
    return result


def pair_motifs(seq_pssm, struct_pssm):
    """Which (sequence motif, structure motif) pairs a combined scan reports.  The reference only ever holds ONE
    motif per side (rnascan.py:262, :298), where combine() (rnascan.py:422-423) joins their tables on
    (Sequence_ID, Start, End).  For multi-PFM libraries (pfmutil.py:89-133):
      * one motif on either side pairs with every motif of the other side (what the join gives);
      * two libraries pair BY MOTIF ID (RNAcompete-S style: one sequence and one structure PFM per protein), by
        position when no id is shared and the libraries have the same size;
    and only pairs of equal width can share (Start, End).  Returns [(seq_id, struct_id)] sorted by the ids, or None
    when two libraries of different size share no id (the caller then joins the two full tables)."""
    sk, tk = list(seq_pssm.keys()), list(struct_pssm.keys())
    if len(sk) == 1 or len(tk) == 1:
        pairs = [(a, b) for a in sk for b in tk]
    else:
        shared = [a for a in sk if a in struct_pssm]
        if shared:
            pairs = [(a, a) for a in shared]
        elif len(sk) == len(tk):
            pairs = list(zip(sk, tk))
        else:
            return None
    return sorted((a, b) for a, b in pairs if seq_pssm[a].length == struct_pssm[b].length)


def scan_combined(engine, records, named_profiles, seq_pssm, struct_pssm, minscore, pairing="aligned",
                  profile_dtype=np.float32, columns=False, prepacked=None):
    """Sequence PFMs + averaged-structure PFMs in ONE pass per PFM width (configs 3 and 5).

    Equivalent to ``combine(scan_main(fasta), scan_main(dir))`` for the records
    that have a profile of the same length under the same Sequence_ID: a window
    is reported for the motif pair (a, b) iff seq_a > minscore AND struct_b > minscore
    (rnascan.py:422-433 is an inner join of two independently thresholded tables); which pairs
    exist is ``pair_motifs``.  Returns None when the inputs cannot be paired one to one (duplicate ids,
    length mismatch, unpairable libraries); callers then take the two-table path.

    ``prepacked`` = (ids, letters, pack.Stream) instead of ``named_profiles``: the profiles of exactly these records,
    in this order, already in stream form (a slice of a packed profile store) -- used as they are when ids and
    lengths agree, no per-record copy."""
    pairs_m = pair_motifs(seq_pssm, struct_pssm)
    if pairs_m is None:
        return None
    if prepacked is not None:
        pids, letters0, pst = prepacked
        batch = _RnaBatch(records)
        if pairs_m and len(batch) and list(batch.ids) == list(pids) and len(set(pids)) == len(pids) and \
                np.array_equal(batch.lengths, pst.lengths):
            prof = pst.profile
            if prof.dtype == np.float64 and np.dtype(profile_dtype) == np.float32:
                # a real downcast (float64 store, float32 rows asked for): one host copy of the batch -- callers keep such
                # batches at one RNASCAN_BATCH_POSITIONS (cli.store_batches)
                prof = np.asarray(prof, dtype=np.float32)
            # a float32 store under float64 rows is scanned AS IT IS: the kernels widen the rows themselves, a float64 copy
            # of the batch (up to 30 GB at 32 x RNASCAN_BATCH_POSITIONS) would change no score and would leave the mapped file
            stream = pack.Stream(batch.codes, prof, batch.offsets, batch.lengths)
            return _scan_combined_stream(engine, stream, batch, list(letters0), pairs_m, seq_pssm, struct_pssm, minscore, pairing,
                                         columns)
        named_profiles = [(pids[r], letters0, pst.profile[int(pst.offsets[r]):int(pst.offsets[r] + pst.lengths[r])])
                          for r in range(len(pids))]
    by_id = {}
    for sid, letters, prof in named_profiles:
        if sid in by_id:
            return None
        by_id[sid] = (letters, prof)
    batch = _RnaBatch(records)
    all_ids = list(batch.ids)
    if len(set(all_ids)) != len(all_ids):
        return None
    keep = [i for i, rid in enumerate(all_ids) if rid in by_id]
    if not keep or not pairs_m:
        return pd.DataFrame(columns=COMBINED_COLUMNS)
    batch = batch.select(np.asarray(keep, dtype=np.int64))
    letters0 = list(by_id[batch.ids[0]][0])
    profs = []
    for k, rid in enumerate(batch.ids):
        letters, prof = by_id[rid]
        if int(batch.lengths[k]) != prof.shape[0] or list(letters) != letters0:
            return None
        profs.append(prof)
    stream = pack.Stream(batch.codes, pack.pack(profiles=profs, profile_dtype=profile_dtype).profile, batch.offsets, batch.lengths)
    return _scan_combined_stream(engine, stream, batch, letters0, pairs_m, seq_pssm, struct_pssm, minscore, pairing, columns)


def scan_pair(engine, seq_records, struct_records, seq_pssm, struct_pssm, minscore, columns=False):
    """Sequence FASTA + structure FASTA in ONE call per motif pair (`rnascan -p .. -q .. seqs.fa structs.fa`,
    rnascan.py:119-123): equivalent to ``combine(scan_main(seqs), scan_main(structs))`` (rnascan.py:416-434) when the
    two batches hold the same records -- same ids in the same order, each id once, same lengths.  The letters of both
    files go to the device as two code streams with the same layout; a window is reported for the motif pair (a, b) iff
    seq_a > minscore AND struct_b > minscore.  Returns None when the batches cannot be paired that way (the caller then
    makes the two tables and joins them)."""
    pairs_m = pair_motifs(seq_pssm, struct_pssm)
    if pairs_m is None:
        return None
    sb, tb = _RnaBatch(seq_records), _RnaBatch(struct_records, fasta.STRUCT)
    ids = list(sb.ids)
    if len(sb) != len(tb) or ids != list(tb.ids) or len(set(ids)) != len(ids) or not np.array_equal(sb.lengths, tb.lengths):
        return None
    if not len(sb) or not pairs_m:
        return pd.DataFrame(columns=COMBINED_COLUMNS)
    stream = pack.Stream(sb.codes, None, sb.offsets, sb.lengths, codes2=tb.codes)
    thr = float(minscore)
    tables = []

    def rows(pos, sq, st, m, seq_ids, struct_ids):
        rec, start = stream.locate(pos)
        lo_seq, lo_st = np.round(sq, 3), _lib.round_decimals(st, 3)       # rnascan.py:273 on a float32 / on a Python float
        return {"_rec": rec, "Sequence_ID": sb.id_column(rec), "Description.Seq": sb.description_column(rec), "Motif_ID.Seq": seq_ids,
                "Start": start + 1, "End": start + m, "Sequence.Seq": sb.windows(pos, m), "LogOdds.Seq": lo_seq,
                "Description.Struct": tb.description_column(rec), "Motif_ID.Struct": struct_ids, "Sequence.Struct": tb.windows(pos, m),
                "LogOdds.Struct": lo_st, "LogOdds.SeqStruct": lo_seq.astype(np.float64) + lo_st}

    by_width = {}
    for a, b in pairs_m:
        by_width.setdefault(seq_pssm[a].length, []).append((a, b))
    for m, group in by_width.items():
        if len(group) > 1 and np.isfinite(thr) and m <= LIBRARY_MAX_M and hasattr(engine, "library_hits_letters"):
            # two libraries (RNAcompete-S style): every pair of this width in ONE pass (k_library over the sequences, the
            # structure letters of its survivors); (position, pair index) order = (record, Start, Motif_ID.Seq, Motif_ID.Struct)
            T = np.stack([seq_pssm[a].letter_table(pack.RNA_LETTERS) for a, _ in group])
            S = np.stack([struct_pssm[b].letter_table(fasta.STRUCT) for _, b in group])
            pos, mo, sq, st = engine.library_hits_letters(stream, T, S, thr, thr)
            tables.append(rows(pos, sq, st, m, table.Indexed([a for a, _ in group], mo), table.Indexed([b for _, b in group], mo)))
            continue
        for a, b in group:
            tab_seq = seq_pssm[a].letter_table(pack.RNA_LETTERS)
            tab_st = struct_pssm[b].letter_table(fasta.STRUCT)
            if np.isneginf(thr) or not hasattr(engine, "hits_pair"):
                # every window with two finite scores is a row: all scores of both sides, the same strict `>` on the host
                sq, _ = engine.scan(stream, tab_seq, None)
                st = engine.scan_letters_f64(pack.Stream(tb.codes, None, tb.offsets, tb.lengths), tab_st)
                pos = np.flatnonzero(stream.window_mask(m) & (sq.astype(np.float64) > thr) & (st > thr))
                sq, st = sq[pos], st[pos]
            else:
                pos, sq, st = engine.hits_pair(stream, tab_seq, tab_st, thr, thr)
            tables.append(rows(pos, sq, st, m, a, b))
    return _finish(tables, COMBINED_COLUMNS, ["Start", "Motif_ID.Seq", "Motif_ID.Struct"], columns)


def _scan_combined_stream(engine, stream, batch, letters0, pairs_m, seq_pssm, struct_pssm, minscore, pairing, columns):
    """the combined hit table of one packed batch (codes + profile rows of the same records)"""
    thr = float(minscore)
    tables = []
    by_width = {}
    for a, b in pairs_m:
        by_width.setdefault(seq_pssm[a].length, []).append((a, b))
    for m, group in by_width.items():
        tabs = [seq_pssm[a].letter_table(pack.RNA_LETTERS) for a, _ in group]
        pssms = [struct_matrix(struct_pssm[b], letters0, pairing) for _, b in group]
        if len(group) > 1 and np.isfinite(thr) and m <= LIBRARY_MAX_M:
            pos, mo, sq, st = engine.library_hits(stream, np.stack(tabs), np.stack(pssms), thr, thr, one_shot=len(by_width) == 1)
            parts = [(pos, mo, sq, st)]
        else:
            parts = []
            for k in range(len(group)):
                pos, sq, st = _select(engine, stream, m, tabs[k], pssms[k], thr, thr, one_shot=len(pairs_m) == 1)
                parts.append((pos, np.full(pos.size, k, dtype=np.int32), sq, st))
        for pos, mo, sq, st in parts:
            rec, start = stream.locate(pos)
            lo_seq = np.round(sq, 3)
            tables.append({
                "_rec": rec,
                "Sequence_ID": batch.id_column(rec),
                "Description.Seq": batch.description_column(rec),
                "Motif_ID.Seq": table.Indexed([a for a, _ in group], mo), "Start": start + 1, "End": start + m,
                "Sequence.Seq": table.Windows(stream.codes, pos, m, pack.RNA_LETTERS),
                "LogOdds.Seq": lo_seq, "Description.Struct": "",
                "Motif_ID.Struct": table.Indexed([b for _, b in group], mo), "Sequence.Struct": ".",
                "LogOdds.Struct": st, "LogOdds.SeqStruct": lo_seq.astype(np.float64) + st})
    # one table: (position, pair index) order = (record, Start, Motif_ID.Seq, Motif_ID.Struct)
    return _finish(tables, COMBINED_COLUMNS, ["Start", "Motif_ID.Seq", "Motif_ID.Struct"], columns)


def _add_match_id(df):
    """rnascan.py:329-332"""
    df["Match_ID"] = list(range(1, df.shape[0] + 1))


def load_profile_dir(directory):
    from . import store
    if store.is_store(directory):
        return store.ProfileStore(directory).named()
    files = fasta.list_profiles(directory)
    parsed = fasta.read_profiles([path for _, path in files])
    return [(sid, letters, prof) for (sid, _), (letters, prof) in zip(files, parsed)]
