"""rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/roofline_probe.py -> profiles/<round>/roofline_pmc_b1convB.json
usage: python tools/pmc_to_json.py gpurun_out/r02b profiles/r02"""
import csv, glob, json, sys
src, dst = sys.argv[1], sys.argv[2]

def per_launch(d, counter):
    path = sorted(glob.glob("%s/%s/*/*counter_collection.csv" % (src, d)))[-1]
    by = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter or "conv_mfma_f6_kernel" not in r["Kernel_Name"]:
            continue
        by[r["Dispatch_Id"]] = by.get(r["Dispatch_Id"], 0.0) + float(r["Counter_Value"])
    vals = sorted(by.values())
    big = [v for v in vals if v > 0.5 * vals[-1]]          # the probe's launches of the dominant layer
    return sum(big) / len(big), len(big)

f, nf = per_launch("pmc_fetch", "FETCH_SIZE")
w, nw = per_launch("pmc_write", "WRITE_SIZE")
rd, wr = f * 1024 * 2, w * 1024
out = {
    "kernel": "conv_mfma_f6_kernel<NT=4> (MPG_PREC_F16F6): resBlock1 convB 5x5 128->128 + 1x1 8->128 shortcut, 8 slices of 256^2, G8 in / G8 out",
    "command": "rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -- python tools/roofline_probe.py 2 6 ; same with --pmc WRITE_SIZE (separate passes; tools/collect_profiles.sh), summarised by tools/pmc_to_json.py",
    "launches_averaged": [nf, nw],
    "FETCH_SIZE_KiB_raw": f, "WRITE_SIZE_KiB": w,
    "correction": "gfx950: FETCH_SIZE counts 64 B per 128-B request on wide coalesced streams -> doubled (MI355X_MICROARCH.md, HBM); WRITE_SIZE exact for 16-B-per-lane stores",
    "hbm_read_bytes": rd, "hbm_write_bytes": wr, "hbm_bytes_per_launch": rd + wr,
    "algorithmic_bytes": {"input_g8_128ch": 268435456, "shortcut_g8_8ch": 16777216, "weights": 2113536, "output_g8_128ch": 268435456},
}
json.dump(out, open(dst + "/roofline_pmc_b1convB.json", "w"), indent=1)
print(json.dumps(out)[:400])
