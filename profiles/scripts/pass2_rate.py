"""Second pass at volume (developer measurement, not the bench line): configs[1]-shaped data, `correct -1` through the executable, second
index at k = 63 (short-read graph coloured by the pass-1 reads), then `correct -2` through the executable with RTK_TRACE kernel times."""
import json, os, subprocess, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
ref_len = int(float(sys.argv[1])) if len(sys.argv) > 1 else 5000000
lr_bases = int(float(sys.argv[2])) if len(sys.argv) > 2 else 64000000
k2 = int(sys.argv[3]) if len(sys.argv) > 3 else 63
extra2 = sys.argv[4:]  # extra options of the second-pass run, e.g. --workers-per-gpu 6 -B 8000000
wd = tempfile.mkdtemp(prefix="rtk_p2_")
t0 = time.time(); pre = bench.make_dataset(wd, ref_len, lr_bases, snps=True); t_data = time.time() - t0
exe = os.path.join(ROOT, "ratatosk_amd", "bin", "Ratatosk")
env = dict(os.environ, RTK_CLI_STATS="1")
t0 = time.time()
r1 = subprocess.run([exe, "correct", "-1", "-c", "16", "-g", pre + ".index.k31.fasta.gz", "-d", pre + ".index.k31.rtsk", "-l", pre + ".lr.fq", "-o", pre], capture_output=True, text=True, env=env)
t_p1 = time.time() - t0
assert r1.returncode == 0, r1.stderr
t0 = time.time()
subprocess.check_call([os.path.join(ROOT, "ratatosk_amd", "bin", "rtk_build_index"), "-s", pre + ".sr.fq", "--colour-reads", pre + ".2.fastq", "-k", str(k2), "-o", pre + ".p2"], stderr=subprocess.DEVNULL)
t_idx2 = time.time() - t0
t0 = time.time()
r2 = subprocess.run([exe, "correct", "-2", "-K", str(k2), "-c", "16", "-g", pre + ".p2.index.k%d.fasta.gz" % k2, "-d", pre + ".p2.index.k%d.rtsk" % k2, "-l", pre + ".2.fastq", "-L", pre + ".lr.fq", "-o", pre] + extra2,
                    capture_output=True, text=True, env=dict(env, RTK_TRACE="1"))
t_p2 = time.time() - t0
assert r2.returncode == 0, r2.stderr
stats = lambda txt: [l for l in txt.splitlines() if "correction phase" in l]
tr = [l for l in r2.stderr.splitlines() if "rtk trace" in l and ("attempt" in l or "phase" in l or "seeds" in l or "shares" in l or "size class" in l or "DFS book" in l)]
# A/B on the same files: RTK_P2_AB="A=1,B=2;C=3" re-runs the second pass once per ';'-separated set of environment overrides
ab = []
for spec in [x for x in os.environ.get("RTK_P2_AB", "").split(";") if x]:
    e2 = dict(env, RTK_TRACE="1", **dict(kv.split("=", 1) for kv in spec.split(",")))
    t0 = time.time(); rr = subprocess.run(r2.args, capture_output=True, text=True, env=e2); ab.append({"env": spec, "rc": rr.returncode, "wall_s": round(time.time() - t0, 2), "pass2": stats(rr.stderr),
               "phase": [l for l in rr.stderr.splitlines() if "rtk trace" in l and "phase" in l][:12]})
print(json.dumps({"ab": ab, "ref_len": ref_len, "lr_bases": lr_bases, "k2": k2, "data_s": round(t_data, 1), "pass1_wall_s": round(t_p1, 2), "pass1": stats(r1.stderr), "index2_s": round(t_idx2, 1),
                  "pass2_wall_s": round(t_p2, 2), "pass2_options": extra2, "pass2": stats(r2.stderr), "pass2_trace_head": tr[:40]}, indent=1))
