"""The reference's own call signatures on top of the MI355X engine.

A maintainer of rnascan can point the names the scan path uses at these and change nothing else:

    rnascan/rnascan.py:258   scan(pssm, seq, alphabet, minscore)
    rnascan/rnascan.py:278   scan_all(seqrecord, pssm, alphabet, minscore)
    rnascan/rnascan.py:293   scan_averaged_structure(struct_file, pssm, minscore)
    rnascan/rnascan.py:335   scan_main(fasta_file, pssm, alphabet, bg, args)      (sequence, directory and SeqRecord branch)
    rnascan/rnascan.py:416   combine(seq_results, struct_results)
    rnascan/rnascan.py:263   pssm.search(seq, threshold=..., both=False)          -> rnascan_amd.pssm.PSSM.search
    matrix.py:68-81          pssm.calculate(sequence)                              -> rnascan_amd.pssm.PSSM.calculate

``pssm`` is the ``{motif_id: PSSM}`` dict ``load_motif`` returns (rnascan.py:210-235), ``alphabet`` either the letters
string (``"GAUC"`` / ``"EHTBLRM"``) or an object with a ``letters`` attribute such as Biopython's alphabets.  Every
function runs on the process-wide default engine (``default_engine()``: one HipEngine on ``RNASCAN_DEVICE``); there is
no CPU path -- without libpfmscan or a gfx950 device the first call raises.
"""
import os

import numpy as np

from . import fasta, scanner

_engine = None


def default_engine():
    """the process-wide HipEngine (created on first use; raises without libpfmscan.so / a gfx950 device)"""
    global _engine
    if _engine is None:
        _engine = scanner.HipEngine(int(os.environ.get("RNASCAN_DEVICE", "0")))
    return _engine


def set_default_engine(engine):
    """use ``engine`` for the functions of this module (tests: an oracle-backed engine; multi-GPU: one per device)"""
    global _engine
    _engine = engine


def _letters(alphabet):
    return alphabet if isinstance(alphabet, str) else alphabet.letters


def _record(seqrecord):
    if isinstance(seqrecord, fasta.Record):
        return seqrecord
    return fasta.Record(seqrecord.id, getattr(seqrecord, "description", ""), str(seqrecord.seq))


def scan(pssm, seq, alphabet, minscore):
    """rnascan.py:258-275: list of [motif_id, Start, End, fragment, round(score, 3)] for one preprocessed sequence"""
    return scanner.scan(default_engine(), pssm, str(seq), _letters(alphabet), minscore)


def scan_all(seqrecord, pssm, alphabet, minscore):
    """rnascan.py:278-286: DataFrame [Motif_ID, Start, End, Sequence, LogOdds] sorted by Start"""
    return scanner.scan_all(default_engine(), _record(seqrecord), pssm, _letters(alphabet), minscore)


def scan_averaged_structure(struct_file, pssm, minscore, pairing="aligned"):
    """rnascan.py:293-315: one averaged-structure profile file -> DataFrame [Motif_ID, Start, End, Sequence, LogOdds].
    ``pairing='positional'`` reproduces the reference's column pairing on Python 3."""
    return scanner.scan_averaged_structure(default_engine(), struct_file, pssm, minscore, pairing, np.float64)


def scan_main(fasta_file, pssm, alphabet, bg, args):
    """rnascan.py:335-413 (``bg`` is unused there as well): FASTA path, averaged-structure directory or a record"""
    from . import cli
    if not hasattr(args, "pairing"):
        args.pairing = "aligned"
    if not hasattr(args, "profile_dtype"):
        args.profile_dtype = "float64"          # the reference computes in fp64 on the values it read
    source = fasta_file if isinstance(fasta_file, str) else _record(fasta_file)
    return cli.scan_main(default_engine(), source, pssm, _letters(alphabet), args)


combine = scanner.combine
_add_match_id = scanner._add_match_id
