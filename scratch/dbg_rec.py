import sys, os, shutil
sys.path.insert(0, os.getcwd())
if len(sys.argv) > 1: shutil.copy(sys.argv[1], "slimfastq_amd/libslimfastq_amd.so")
from slimfastq_amd import capi
fq = capi.synth_fastq(300, 150, 1, 0)
print(fq[:70])
ctx = capi.Context()
enc = ctx.encode_host(fq, level=3, block_reads=128, prior_step=1, tables=capi.TABLES_FROZEN, chain_reads=64)
print("done", len(enc.stream("rec")))
