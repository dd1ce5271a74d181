"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/roofline_probe_wgrad.py -> profiles/<round>/roofline_pmc_wgrad.json
usage: python tools/pmc_wgrad_to_json.py gpurun_out/r02c profiles/r02 <calls>"""
import csv, glob, json, sys
src, dst, calls = sys.argv[1], sys.argv[2], int(sys.argv[3])

def table(d, counter):
    path = sorted(glob.glob("%s/%s/*/*counter_collection.csv" % (src, d)))[-1]
    by = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "")
        k = k.split("(")[0] if not k.startswith("_Z") else k
        e = by.setdefault(k, {})
        e[r["Dispatch_Id"]] = e.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
    return {k: (sum(v.values()), len(v)) for k, v in by.items()}

f, w = table("pmc_wgrad_fetch", "FETCH_SIZE"), table("pmc_wgrad_write", "WRITE_SIZE")
out = {"call": "the weight gradient as the training step issues it: 5x5 128->128, 16 tiles of 256^2; x = the forward launch's G8 operand "
               "(512 MiB, no pass of its own), dy fp32 NHWC 512 MiB -> G8 scaled by max |dy| (one conversion shared with the data "
               "gradient, listed), mpg_conv2d_wgrad_g8 MPG_PREC_F16X3 (wgrad_ring_kernel<5,5,1,3>)",
       "command": "rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python tools/roofline_probe_wgrad.py %d ; same with "
                  "--pmc WRITE_SIZE (separate passes); summarised by tools/pmc_wgrad_to_json.py" % calls,
       "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request on wide coalesced streams -> doubled; WRITE_SIZE as counted",
       "per_kernel_per_call": {}, "calls": calls}
tot = 0.0
for k in sorted(set(f) | set(w)):
    if not any(t in k for t in ("absmax", "to_g8", "wgrad_")):
        continue
    rd = f.get(k, (0, 0))[0] * 1024 * 2 / calls
    wr = w.get(k, (0, 0))[0] * 1024 / calls
    out["per_kernel_per_call"][k] = {"launches_per_call": f.get(k, w.get(k))[1] / calls, "hbm_read_bytes": rd, "hbm_write_bytes": wr}
    tot += rd + wr
out["hbm_bytes_per_call"] = tot
out["algorithmic_bytes_per_call"] = {"x_g8": 536870912, "dy_g8": 536870912, "dw": 1638400}
json.dump(out, open(dst + "/roofline_pmc_wgrad.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:1800])
