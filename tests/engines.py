"""TEST-ONLY engine: same interface as rnascan_amd.scanner.HipEngine, scores come from
the CPU oracle.  Lets the host-side table logic be checked on a GPU-less machine; the
product package never imports this."""
import numpy as np

from oracle import oracle


class OracleEngine(object):
    def scan(self, stream, letter_table=None, struct_pssm=None):
        sq = oracle.stream_seq(stream.codes, letter_table) if letter_table is not None else None
        st = oracle.stream_struct(stream.profile, struct_pssm) if struct_pssm is not None else None
        return sq, st

    def pwm_calculate(self, sequence, matrix):
        return oracle.pwm_calculate(sequence, matrix)

    def scan_letters_f64(self, stream, letter_table):
        return oracle.stream_letters_f64(stream.codes, letter_table)

    def hits_letters_f64(self, stream, letter_table, thr):
        """fp64 letter scores above thr (matrix.py:25-43 + the strict `>` of rnascan.py:263) -> (pos, score)"""
        sc = oracle.stream_letters_f64(stream.codes, letter_table)
        pos = oracle.stream_hits(None, sc, -np.inf, thr)
        return pos, sc[pos]

    def hits_pair(self, stream, seq_table, struct_table, thr_seq, thr_struct):
        """two code streams: float32 sequence score AND fp64 structure-letter score above their thresholds"""
        sq = oracle.stream_seq(stream.codes, seq_table)
        st = oracle.stream_letters_f64(stream.codes2, struct_table)
        pos = oracle.stream_hits(sq, st, thr_seq, thr_struct)
        return pos, sq[pos], st[pos]

    def hits(self, stream, letter_table=None, struct_pssm=None, thr_seq=-np.inf, thr_struct=-np.inf, one_shot=True):
        sq, st = self.scan(stream, letter_table, struct_pssm)
        pos = oracle.stream_hits(sq, st, thr_seq, thr_struct)
        return pos, (None if sq is None else sq[pos]), (None if st is None else st[pos])

    def library_hits(self, stream, letter_tables, struct_pssms, thr_seq, thr_struct=None, one_shot=True):
        """per-motif oracle scans -> (pos, motif, seq, struct | None) sorted by (pos, motif), like HipEngine.library_hits"""
        n = (letter_tables if letter_tables is not None else struct_pssms).shape[0]
        if letter_tables is None:                            # structure-only library
            tt = np.broadcast_to(np.asarray(thr_struct, dtype=np.float64), (n,))
            pos, mo, st_l = [], [], []
            for k in range(n):
                st = oracle.stream_struct(stream.profile, struct_pssms[k])
                p = oracle.stream_hits(None, st, -np.inf, tt[k])
                pos.append(p)
                mo.append(np.full(p.size, k, dtype=np.int32))
                st_l.append(st[p])
            pos, mo, st_l = np.concatenate(pos), np.concatenate(mo), np.concatenate(st_l)
            order = np.lexsort((mo, pos))
            return pos[order], mo[order], None, st_l[order]
        ts = np.broadcast_to(np.asarray(thr_seq, dtype=np.float64), (n,))
        tt = np.broadcast_to(np.asarray(-np.inf if thr_struct is None else thr_struct, dtype=np.float64), (n,))
        pos, mo, sq_l, st_l = [], [], [], []
        for k in range(n):
            sq = oracle.stream_seq(stream.codes, letter_tables[k])
            st = oracle.stream_struct(stream.profile, struct_pssms[k]) if struct_pssms is not None else None
            p = oracle.stream_hits(sq, st, ts[k], tt[k] if st is not None else -np.inf)
            pos.append(p)
            mo.append(np.full(p.size, k, dtype=np.int32))
            sq_l.append(sq[p])
            st_l.append(st[p] if st is not None else np.zeros(p.size))
        pos, mo, sq_l, st_l = np.concatenate(pos), np.concatenate(mo), np.concatenate(sq_l), np.concatenate(st_l)
        order = np.lexsort((mo, pos))
        return pos[order], mo[order], sq_l[order], (st_l[order] if struct_pssms is not None else None)

    def close(self):
        pass
