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

    def scan_letters_f64(self, stream, letter_table):
        return oracle.stream_letters_f64(stream.codes, letter_table)

    def hits(self, stream, letter_table=None, struct_pssm=None, thr_seq=-np.inf, thr_struct=-np.inf):
        sq, st = self.scan(stream, letter_table, struct_pssm)
        pos = oracle.stream_hits(sq, st, thr_seq, thr_struct)
        return pos, (None if sq is None else sq[pos]), (None if st is None else st[pos])

    def close(self):
        pass
