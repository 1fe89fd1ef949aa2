for rep in 1 2; do
for lib in tools/variants/libaukit_oldrows.so aukit_amd/libaukit_hip.so; do
rm -rf /tmp/dl; PYTHONPATH=$PWD AUKIT_LIB=$PWD/$lib timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/dl -o dl -- python3 /tmp/dl.py > /dev/null 2>&1
python3 - $lib <<'PY'
import csv, glob, sys
bm, ch = [], []
for f in glob.glob("/tmp/dl/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        d = round((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, 2)
        if "blockmaps" in r["Kernel_Name"]: bm.append(d)
        if "k_df_chunks" in r["Kernel_Name"]: ch.append(d)
print(sys.argv[1][-16:], "maps", bm, "chunks", ch)
PY
done; done
