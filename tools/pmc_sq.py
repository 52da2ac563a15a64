"""rocprofv3 --pmc counter CSVs -> one JSON: mean counter value per launch for every kernel whose name matches a filter.

    python tools/pmc_sq.py <out.json> <filter-substring> <dir> [<dir> ...]

Collect each directory in its own rocprofv3 pass (8 SQ counters fit one pass; --kernel-trace only, no other trace domain):
see tools/gpu_pmc_igemm128.sh.  Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles
summed over waves, SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs, SQ_BUSY_CYCLES is per shader engine."""
import csv
import glob
import json
import sys
from collections import defaultdict

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from pmc_traffic import short  # noqa: E402


def main():
    out, filt, dirs = sys.argv[1], sys.argv[2], sys.argv[3:]
    tot, cnt = defaultdict(lambda: defaultdict(float)), defaultdict(lambda: defaultdict(int))
    for d in dirs:
        for path in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
            with open(path) as f:
                for row in csv.DictReader(f):
                    k = short(row["Kernel_Name"])
                    if filt in k:
                        tot[k][row["Counter_Name"]] += float(row["Counter_Value"])
                        cnt[k][row["Counter_Name"]] += 1
    doc = {}
    for k in tot:
        m = {c: tot[k][c] / max(cnt[k][c], 1) for c in sorted(tot[k])}
        derived = {}
        if m.get("SQ_WAVE_CYCLES"):
            wc = m["SQ_WAVE_CYCLES"]
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_ANY"):
                if c in m:
                    derived[c + "/SQ_WAVE_CYCLES"] = round(m[c] / wc, 4)
        if m.get("SQ_LDS_IDX_ACTIVE") and "SQ_LDS_BANK_CONFLICT" in m:
            derived["SQ_LDS_BANK_CONFLICT/SQ_LDS_IDX_ACTIVE"] = round(m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"], 4)
        if m.get("SQ_BUSY_CYCLES") and m.get("SQ_VALU_MFMA_BUSY_CYCLES"):
            # SQ_BUSY_CYCLES is summed over the 32 shader engines (8 XCDs x 4), SQ_VALU_MFMA_BUSY_CYCLES over the 1024 SIMDs
            derived["mfma_pipe_utilisation"] = round(m["SQ_VALU_MFMA_BUSY_CYCLES"] / (m["SQ_BUSY_CYCLES"] / 32.0 * 1024.0), 4)
        if m.get("SQ_INSTS_MFMA") and m.get("SQ_INSTS_VALU"):
            derived["valu_insts_per_mfma"] = round((m["SQ_INSTS_VALU"] - m["SQ_INSTS_MFMA"]) / m["SQ_INSTS_MFMA"], 3)
        doc[k] = {"launches": max(cnt[k].values()), "mean_per_launch": {c: round(v) for c, v in m.items()}, "ratios": derived}
    with open(out, "w") as f:
        json.dump(doc, f, indent=1)
    print(json.dumps(doc, indent=1))


if __name__ == "__main__":
    main()
