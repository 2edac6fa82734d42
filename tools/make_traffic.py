#!/usr/bin/env python3
"""profiles/traffic.json from the PMC passes of tools/profile_round.sh.

    python tools/make_traffic.py <dir> <tag>=<key> ...      e.g.  gpurun_out r03_f32=f32:64x512x512:1024

For every tag reads <dir>/<tag>_pmc_FETCH_SIZE.md and _pmc_WRITE_SIZE.md, takes the search kernel's row (the full template
instance as rocprofv3 prints it) and writes the fabric-side bytes per launch under the key bench.py looks up
(dtype:TxHxW:candidates[:sigmag]); bench.py attaches an entry only to a run that launched the same instance."""
import json
import os
import re
import sys


def search_row(path, counter):
    for line in open(path):
        m = re.match(r"\| `(kb::kb_search_[^`]*)` \| %s \| (\d+) \| ([0-9.e+]+) \|" % counter, line)
        if m:
            return m.group(1).replace("(kb::SearchArgs)", ""), int(m.group(2)), float(m.group(3))
    raise SystemExit(f"no search kernel row with {counter} in {path}")


def main():
    d = sys.argv[1]
    table = {"_comment": "HBM-side (fabric) bytes per search-kernel launch from rocprofv3 PMC passes (tools/profile_round.sh: separate "
                         "--pmc passes for FETCH_SIZE and WRITE_SIZE, units KiB).  gfx950 correction: FETCH_SIZE counts 128-byte "
                         "requests as 64 bytes (calibrated on kb_pad_kernel<4,true>, which reads the array exactly once); WRITE_SIZE "
                         "needs none.  bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.  Infinity-Cache hits are included in these "
                         "counters: for arrays below 256 MiB they bound DRAM traffic from above.  Keys: dtype:TxHxW:candidates[:sigmag]; "
                         "kernel_instance = the template instance the numbers belong to (bench.py rejects the entry for any other)."}
    for spec in sys.argv[2:]:
        tag, key = spec.split("=", 1)
        fpath, wpath = (os.path.join(d, f"{tag}_pmc_{c}.md") for c in ("FETCH_SIZE", "WRITE_SIZE"))
        name, n, fetch = search_row(fpath, "FETCH_SIZE")
        name_w, _, write = search_row(wpath, "WRITE_SIZE")
        assert name == name_w, (name, name_w)
        table[key] = {"kernel_instance": name, "fetch_kib": fetch, "write_kib": write,
                      "bytes": int((2 * fetch + write) * 1024), "dispatches": n,
                      "source": f"profiles/{tag}_pmc_FETCH_SIZE.md, profiles/{tag}_pmc_WRITE_SIZE.md"}
    json.dump(table, sys.stdout, indent=2)
    print()


if __name__ == "__main__":
    main()
