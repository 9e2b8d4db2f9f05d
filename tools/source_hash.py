"""sha256 over the sources libfaqcs_mi.so is built from (faqcs_amd/csrc/*.hip, *.h, include/*.h): what ties a stored counter file
(profiles/traffic_*.json, *_counters.json) to the build it was measured on.  `python tools/source_hash.py` prints it.
faqcs_pargz.h is not among them: it is host code of the command line (faqcs_cli.cpp includes it, the library does not)."""
import glob
import hashlib
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST_ONLY = {"faqcs_pargz.h"}  # headers under csrc/ that only faqcs_cli.cpp includes


def source_hash():
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "faqcs_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "faqcs_amd", "csrc", "*.h"))
                   + glob.glob(os.path.join(ROOT, "include", "*.h")))
    files = [f for f in files if os.path.basename(f) not in HOST_ONLY]
    for f in files:
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()


def git_head():
    try:
        return subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None
    except Exception:
        return None


def stamp(d):
    """Adds the build identity to a counters dictionary."""
    d["source_sha256"] = source_hash()
    d["git_head_when_measured"] = git_head()  # (None on the GPU box: the snapshot has no .git)
    return d


def load_if_current(path):
    """(counters, None) when the file was measured on the sources in this tree, else (None, why)."""
    import json

    if not os.path.exists(path):
        return None, "%s does not exist" % os.path.relpath(path, ROOT)
    try:
        d = json.load(open(path))
    except Exception as e:
        return None, "%s is unreadable (%s)" % (os.path.relpath(path, ROOT), e)
    if d.get("source_sha256") != source_hash():
        return None, ("%s was measured on other kernel sources (its source_sha256 %s..., this tree %s...): re-run profiles/collect_r6.sh"
                      % (os.path.relpath(path, ROOT), str(d.get("source_sha256"))[:10], source_hash()[:10]))
    return d, None


if __name__ == "__main__":
    print(source_hash())
