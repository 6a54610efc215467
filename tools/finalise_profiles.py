"""Developer tool (this container, after a gpurun call that left bench_final.json, phase_light.txt, phase_full.txt, soak.txt, sweep.txt under gpurun_out/): copy them into
profiles/<tag>_* with their headers.  The rocprofv3 summaries are made by tools/summarise_profiles.py / summarise_wait_profiles.py.   python tools/finalise_profiles.py r04"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"
G = lambda n: os.path.join(ROOT, "gpurun_out", n)
Pf = lambda n: os.path.join(ROOT, "profiles", f"{tag}_{n}")

json.dump(json.load(open(G("bench_final.json"))), open(Pf("bench_line.json"), "w"), indent=1)

def keep_header(path, body):
    """the leading '#' lines of the committed file stay (they describe what the file is); the measured lines below them are replaced"""
    head = []
    for line in open(path):
        if not line.startswith("#"):
            break
        head.append(line)
    open(path, "w").write("".join(head) + body)

keep_header(Pf("phase_profile.txt"), "## LIGHT\n" + open(G("phase_light.txt")).read() + "## FULL\n" + open(G("phase_full.txt")).read())
keep_header(Pf("soak.txt"), open(G("soak.txt")).read())
old = open(Pf("accuracy_sweep.txt")).read()
i = old.index("#\n# Choice of the N > 20 default tolerance")
head = "".join(l for l in old[:i].splitlines(True) if l.startswith("#"))
open(Pf("accuracy_sweep.txt"), "w").write(head + open(G("sweep.txt")).read() + old[i:])
print("written:", Pf("bench_line.json"), Pf("phase_profile.txt"), Pf("soak.txt"), Pf("accuracy_sweep.txt"))
