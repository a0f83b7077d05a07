#!/usr/bin/env python3
"""ONE script for everything bench.py reads back from profiles/: regenerates profiles/committed_kernel_times.json and
profiles/pmc_traffic.json from a tools/collect_profiles.sh run (gpurun_out/<tag>_*), and stamps every entry with the
sha256 of the kernel source file(s) it was measured on -- bench.py recomputes those hashes at run time and DROPS an entry
whose kernel has changed since (there is no git on the GPU box: the file contents are the identity).
    python tools/make_committed.py <tag>          (after copying gpurun_out/<tag>_* into profiles/)"""
import csv, hashlib, json, os, re, sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(root, "coarsegrainingvae_amd", "csrc")
# kernel family -> the source files that define it (what a stale entry is checked against)
SOURCES = {
    "equi_msg_fwd": ["equi_msg_grp.hip", "equi_msg.hip", "equi_msg_dev.h"],
    "segment_reduce": ["scatter.hip"],
    "optimizer": ["optim.hip"],
    "rank_update": ["skinny_gemm.hip"],
}


def source_hash(files):
    h = hashlib.sha256()
    for f in files:
        h.update(open(os.path.join(CSRC, f), "rb").read())
    return h.hexdigest()[:16]


def src(path):
    for d in ("profiles", "gpurun_out"):
        p = os.path.join(root, d, path)
        if os.path.exists(p):
            return p
    return None


GROUPS = [
    ("decoder layer (channel-group kernels incl. the 4-column-block Dense forward, which the prior / heads share)", r"dec_|skinny_fwd_k<1, 16>"),
    ("optimizer (rank update of the bead-level layers, norm, Adam)", r"rank_update_mixed_k|grouped_wgrad_t<true>|adam_update|sumsq_partial|optim_finalize|wgrad_gram"),
    ("atom-graph message passing (K2g / K2 / K2b + reductions)", r"equi_msg_|segment_reduce|segment_broadcast"),
    ("atom-level Dense (tile GEMMs) and their weight gradients", r"tile_|gathered_wgrad|grouped_wgrad_t<false>|wgrad_split"),
    ("per-batch graph plans + edge records", r"pj_|grp_build|gj_records|csr_|edge_geometry|copyBuffer|fillBuffer|batch_rows"),
    ("prior / heads / loss / decoder tail / bead-level blocks", r"skinny_|elbo|loss_tail|reconstruct|embedding_rows|pseudo_|update_|reparam|prior_msg"),
]


def pmc_means(path):
    """kernel symbol -> mean counter value per dispatch (tools/pmc_summary.py output)."""
    out = {}
    if not path:
        return out
    for line in open(path):
        m = re.match(r"(.*?) \| (\w+): n=(\d+) mean=([0-9.]+)(?: max=([0-9.]+) med=([0-9.]+))?", line)
        if m:                                             # (median dispatch where the summary has it, count, largest dispatch)
            out[m.group(1).strip()] = (float(m.group(6) or m.group(4)), int(m.group(3)), float(m.group(5) or m.group(4)))
    return out


times, traffic = {}, {"_comment": (
    "HBM-side traffic per launch from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, eager launches of "
    "`bench.py --no-graph`: the kernel runs between the other kernels of the step, caches as in training). FETCH_SIZE is in "
    "KiB and on gfx950 reports HALF the bytes of a coalesced read whatever the width per lane -- 4, 8 or 16 bytes -- and of "
    "8-byte row gathers too (tools/probes/fetch_calib.hip, profiles/%s_fetch_calibration.txt: 1 GiB streamed -> 524 29x KiB): "
    "every fetch figure is doubled. WRITE_SIZE is exact (1 GiB memset -> 1 048 576 KiB). Each entry carries the sha256 of the "
    "kernel sources it was measured on; bench.py drops entries whose sources have changed." % tag)}
for w in ("chignolin", "dipeptide", "protein2000"):
    entry, tr = {}, {}
    stats = src(f"{tag}_{w}_kernel_stats.csv")
    bench = src(f"{tag}_bench_{w}.json")
    line = None
    if bench:
        try:
            line = json.loads([l for l in open(bench) if l.startswith("{")][-1])
        except (ValueError, IndexError):
            line = None
    if stats:
        rows = list(csv.DictReader(open(stats)))
        fwd = [r for r in rows if "equi_msg_fwd_grp_k" in r["Name"]] or [r for r in rows if "equi_msg_fwd_k" in r["Name"]]
        if fwd:
            r = max(fwd, key=lambda r: int(r["TotalDurationNs"]))
            entry["message_forward"] = {
                "kernel": r["Name"].split("(")[0].replace("void ", ""), "avg_us": float(r["AverageNs"]) / 1e3, "calls": int(r["Calls"]),
                "source": f"profiles/{tag}_{w}_kernel_stats.csv (rocprofv3 --kernel-trace --stats of `python bench.py --workload {w} --no-cpu-baseline --no-parity`)",
                "source_sha256": source_hash(SOURCES["equi_msg_fwd"]), "source_files": SOURCES["equi_msg_fwd"]}
    seq = src(f"{tag}_step_sequence_{w}.txt")
    if seq:
        tot, per, other = 0.0, {g: [0.0, 0] for g, _ in GROUPS}, [0.0, 0]
        for l in open(seq):
            m = re.search(r"dur\s+([0-9.]+)\s+(?:grid|blocks)=.*?x\s*\d+\s+(.*)$", l)
            if not m:
                continue
            dur, name = float(m.group(1)), m.group(2)
            tot += dur
            for g, pat in GROUPS:
                if re.search(pat, name):
                    per[g][0] += dur; per[g][1] += 1
                    break
            else:
                other[0] += dur; other[1] += 1
        groups = [{"group": g, "us": round(v[0], 1), "launches": v[1], "share": round(v[0] / tot, 3)} for g, v in per.items()]
        groups.append({"group": "unclassified", "us": round(other[0], 1), "launches": other[1], "share": round(other[0] / tot, 3)})
        entry["kernel_groups"] = {"source": f"profiles/{tag}_step_sequence_{w}.txt (one replayed step incl. the per-batch graph work, rocprofv3 kernel trace)",
                                  "kernel_time_us": round(tot, 1), "groups": sorted(groups, key=lambda g: -g["us"])}
    if entry:
        times[w] = entry
    # ---- PMC traffic
    fetch, write = pmc_means(src(f"{tag}_pmc_FETCH_SIZE_{w}.txt")), pmc_means(src(f"{tag}_pmc_WRITE_SIZE_{w}.txt"))

    def pick(pattern, largest=False):
        ks = [k for k in fetch if re.search(pattern, k)]
        if not ks:
            return None
        k = max(ks, key=lambda k: fetch[k][0] * fetch[k][1])
        i = 2 if largest else 0
        return k, fetch[k][i], write.get(k, (0.0, 0, 0.0))[i]
    rk = "message_forward"                                # bench.py: roofline["pmc_key"]
    hit = pick(r"equi_msg_fwd_grp_k|equi_msg_fwd_k")
    if rk and hit:
        sym, f_kib, w_kib = hit
        tr[rk] = {"fetch_KiB": f_kib, "fetch_factor": 2, "write_KiB": w_kib, "traffic_bytes": int(1024 * (2 * f_kib + w_kib)),
                  "kernel_symbol": sym.split("(")[0].replace("void ", ""), "source": f"profiles/{tag}_pmc_*_{w}.txt",
                  "source_sha256": source_hash(SOURCES["equi_msg_fwd"]), "source_files": SOURCES["equi_msg_fwd"]}
    hit = pick(r"segment_reduce_k", largest=True)
    if hit:
        tr["segment_reduce_k<4,256,8>"] = {"fetch_KiB": hit[1], "fetch_factor": 2, "write_KiB": hit[2], "traffic_bytes": int(1024 * (2 * hit[1] + hit[2])),
                                           "note": "LARGEST segment_reduce dispatch of the run = the standalone [E,F,3] -> [N,F,3] reduction bench.py times as `scatter_add` (the PMC passes run bench.py with its extras on)",
                                           "source": f"profiles/{tag}_pmc_*_{w}.txt",
                                           "source_sha256": source_hash(SOURCES["segment_reduce"]), "source_files": SOURCES["segment_reduce"]}
    rank_kernel = "rank_update_mixed_k" if pick(r"rank_update_mixed_k") else "grouped_wgrad_t<true>"
    hit = pick(re.escape(rank_kernel))
    if hit:
        tr[rank_kernel] = {"fetch_KiB": hit[1], "fetch_factor": 2, "write_KiB": hit[2], "traffic_bytes": int(1024 * (2 * hit[1] + hit[2])),
                                       "source": f"profiles/{tag}_pmc_*_{w}.txt", "source_sha256": source_hash(SOURCES["rank_update"]),
                                       "source_files": SOURCES["rank_update"]}
    opt = [pick(p) for p in (r"sumsq_partial", r"optim_finalize", r"adam_update")]
    rank = [pick(p) for p in (r"wgrad_gram_k", r"wgrad_gram_reduce_k", re.escape(rank_kernel))]
    if all(opt):
        f_kib, w_kib = sum(o[1] for o in opt), sum(o[2] for o in opt)
        key, files = "sumsq_partial+optim_finalize+adam_update", SOURCES["optimizer"]
        if all(rank):                                     # the rank-update step (bench.py: optimizer_roofline's kernel name)
            f_kib += sum(o[1] for o in rank); w_kib += sum(o[2] for o in rank)
            key, files = "wgrad_gram+optim_finalize+adam_update+" + rank_kernel, SOURCES["optimizer"] + SOURCES["rank_update"]
        tr[key] = {"fetch_KiB": f_kib, "fetch_factor": 2, "write_KiB": w_kib, "traffic_bytes": int(1024 * (2 * f_kib + w_kib)),
                   "note": "median dispatch of each kernel of the optimiser step, summed", "source": f"profiles/{tag}_pmc_*_{w}.txt",
                   "source_sha256": source_hash(files), "source_files": files}
    if tr:
        traffic[w] = tr
json.dump(times, open(os.path.join(root, "profiles", "committed_kernel_times.json"), "w"), indent=1)
json.dump(traffic, open(os.path.join(root, "profiles", "pmc_traffic.json"), "w"), indent=1)
print(json.dumps({w: list(v) for w, v in traffic.items() if w != "_comment"}, indent=1))
print(json.dumps({w: {k: (v.get("avg_us") if isinstance(v, dict) else None) for k, v in e.items()} for w, e in times.items()}, indent=1))
