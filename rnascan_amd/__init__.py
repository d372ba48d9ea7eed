"""MI355X-native sliding-window PFM scanner behind rnascan's interfaces.

_lib     ctypes binding of libpfmscan.so (include/pfmscan.h) -- the only thing that computes scores
pack     packed record stream (one separator position after every record)
pssm     PFM files -> log-odds operands
fasta    FASTA / background / averaged-structure files
scanner  host-side mirror of rnascan's scan layer (batch scans, fused combined scan)
shard    record sharding over the GPUs of a node (one process per GPU)
store    packed, memory-mapped averaged-structure profile store
table    streaming TSV writer
cli      the `rnascan` command line
"""
__version__ = "0.10.2+mi355x.1"
