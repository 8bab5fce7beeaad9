"""slimfastq-amd.multi -- compress / decompress many FASTQ files across the GPUs of a node.

The role of the reference's tools/slimfastq.multi (a Perl script that keeps N `slimfastq` processes busy
over the files of some directories), with the same options where they apply.  Here the unit of
parallelism is the GPU, not the CPU core: one `slimfastq-amd -b` worker process per GPU keeps one
library context (HIP runtime, model tables) alive and receives its files one at a time over a pipe, so
the files balance dynamically and nothing is re-initialised per file.  A second worker per GPU (-c 2 x GPUs)
overlaps one worker's file I/O with the other's kernels.

    python -m slimfastq_amd.multi [OPTIONS] directories/files
      -d, --decompress        decompress .sfq files (default: compress)
      -r, --recursively       descend into directories
      -t, --tgt_dir DIR       write results there (default: next to each source)
      -f, --fq_suffix LIST    comma separated, default '.fastq,.fq' (decompression writes the first)
      -s, --sfq_suffix SFX    default '.sfq'
      -c, --count N           worker processes (default: one per GPU)
      -g, --gpus LIST         HIP device indices to use, default: all visible
      -e, --exec PATH         slimfastq-amd executable (default: slimfastq_amd/bin/slimfastq-amd)
      -l, --level N           compression level 1..4 (default 3)
      -B, --block_reads N     records per block (default: automatic, about 376 KiB of text per block)
      -O, --overwrite         replace existing targets (default: skip them, like the reference script)
      -v, --verbose
"""
import argparse
import os
import queue
import subprocess
import sys
import threading
import time


def find_files(args, suffixes):
    out = []
    for a in args.paths:
        if os.path.isdir(a):
            for root, dirs, files in os.walk(a):
                dirs.sort()
                for f in sorted(files):
                    if f.endswith(tuple(suffixes)):
                        out.append(os.path.join(root, f))
                if not args.recursively:
                    break
        elif os.path.exists(a):
            out.append(a)
        else:
            print("%s: no such file or directory" % a, file=sys.stderr)
    return out


def target_of(src, args, suffixes, new_suffix):
    base = os.path.basename(src)
    for s in suffixes:
        if base.endswith(s):
            base = base[: -len(s)]
            break
    d = args.tgt_dir or os.path.dirname(src) or "."
    return os.path.join(d, base + new_suffix)


def visible_gpus():
    try:
        import torch
        n = torch.cuda.device_count()        # does not initialise the GPU
        if n:
            return list(range(n))
    except Exception:
        pass
    return [0]


class Worker(threading.Thread):
    """One `slimfastq-amd -b` process; jobs come from the shared queue, one line in, one line out."""

    def __init__(self, wid, gpu, cmd, jobs, results, verbose):
        super().__init__(daemon=True)
        self.wid, self.gpu, self.cmd, self.jobs, self.results, self.verbose = wid, gpu, cmd, jobs, results, verbose

    def run(self):
        proc = subprocess.Popen(self.cmd + ["-g", str(self.gpu)], stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True, bufsize=1)
        try:
            while True:
                try:
                    src, dst = self.jobs.get_nowait()
                except queue.Empty:
                    break
                t0 = time.time()
                try:
                    proc.stdin.write("%s\t%s\n" % (src, dst))
                    proc.stdin.flush()
                    ans = proc.stdout.readline()
                except (BrokenPipeError, OSError):
                    ans = ""
                if not ans:                     # the worker died (no GPU, crashed): its job and the rest go back
                    self.results.append((src, dst, False, "worker on GPU %d exited" % self.gpu, 0.0))
                    break
                f = ans.rstrip("\n").split("\t")
                ok = f[0] == "ok"
                self.results.append((src, dst, ok, "" if ok else (f[2] if len(f) > 2 else ans.strip()), time.time() - t0))
                if self.verbose:
                    print("[gpu %d] %s %s -> %s (%.2fs)%s" % (self.gpu, "ok  " if ok else "FAIL", src, dst, time.time() - t0,
                                                             "" if ok else ": " + self.results[-1][3]), flush=True)
        finally:
            try:
                proc.stdin.close()
            except OSError:
                pass
            proc.wait()


def main(argv=None):
    ap = argparse.ArgumentParser(prog="slimfastq-amd.multi", description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("paths", nargs="+")
    ap.add_argument("-d", "--decompress", action="store_true")
    ap.add_argument("-r", "--recursively", action="store_true")
    ap.add_argument("-t", "--tgt_dir")
    ap.add_argument("-f", "--fq_suffix", default=".fastq,.fq")
    ap.add_argument("-s", "--sfq_suffix", default=".sfq")
    ap.add_argument("-c", "--count", type=int, default=0)
    ap.add_argument("-g", "--gpus", default="")
    ap.add_argument("-e", "--exec", dest="exe", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "bin", "slimfastq-amd"))
    ap.add_argument("-l", "--level", type=int, default=3)
    ap.add_argument("-B", "--block_reads", type=int, default=-1)
    ap.add_argument("-O", "--overwrite", action="store_true")
    ap.add_argument("-v", "--verbose", action="store_true")
    args = ap.parse_args(argv)

    if not os.access(args.exe, os.X_OK):
        print("Can't find a slimfastq-amd executable at %s (build it with `python -m slimfastq_amd.build`, or use -e)" % args.exe, file=sys.stderr)
        return 1
    fq_suffixes = [s.strip() for s in args.fq_suffix.split(",") if s.strip()] or [".fastq", ".fq"]
    if args.tgt_dir and not os.path.isdir(args.tgt_dir):
        os.mkdir(args.tgt_dir)                      # one level only, like the reference script
    if args.decompress:
        sources = find_files(args, [args.sfq_suffix])
        pairs = [(s, target_of(s, args, [args.sfq_suffix], fq_suffixes[0])) for s in sources]
    else:
        sources = find_files(args, fq_suffixes)
        pairs = [(s, target_of(s, args, fq_suffixes, args.sfq_suffix)) for s in sources]
    if not pairs:
        print("Missing directories or files to %s" % ("decompress" if args.decompress else "compress"), file=sys.stderr)
        return 1
    jobs = queue.Queue()
    skipped = 0
    # largest first: the long files start early, the short ones fill the gaps
    for src, dst in sorted(pairs, key=lambda p: -os.path.getsize(p[0])):
        if os.path.exists(dst) and not args.overwrite:
            skipped += 1
            if args.verbose:
                print("skip %s: %s exists" % (src, dst))
            continue
        jobs.put((src, dst))
    gpus = [int(g) for g in args.gpus.split(",") if g.strip() != ""] or visible_gpus()
    nworkers = max(1, min(args.count or len(gpus), jobs.qsize()))
    cmd = [args.exe, "-b", "-l", str(args.level)] + (["-B", str(args.block_reads)] if args.block_reads >= 0 else []) + \
          (["-d"] if args.decompress else []) + (["-O"] if args.overwrite else [])
    per_gpu = (nworkers + len(gpus) - 1) // len(gpus)
    if per_gpu > 1:                                  # workers sharing a GPU share its memory: model tables take most of it
        cmd += ["-T", str(max(5, 60 // per_gpu))]
    results = []
    t0 = time.time()
    workers = [Worker(i, gpus[i % len(gpus)], cmd, jobs, results, args.verbose) for i in range(nworkers)]
    for w in workers:
        w.start()
    for w in workers:
        w.join()
    left = jobs.qsize()
    bad = [r for r in results if not r[2]]
    done = [r for r in results if r[2]]
    nbytes = sum(os.path.getsize(r[0]) for r in done if os.path.exists(r[0]))
    dt = time.time() - t0
    print("%d file(s) %s in %.1fs on %d worker(s) / %d GPU(s) (%.1f MB/s of source bytes); %d skipped, %d failed, %d not attempted" %
          (len(done), "decompressed" if args.decompress else "compressed", dt, nworkers, len(set(gpus[i % len(gpus)] for i in range(nworkers))),
           nbytes / 1e6 / max(dt, 1e-9), skipped, len(bad), left))
    for r in bad:
        print("FAILED %s: %s" % (r[0], r[3]), file=sys.stderr)
    return 0 if not bad and not left else 2


if __name__ == "__main__":
    sys.exit(main())
