"""Print the kernel timeline of one train step from a rocprofv3 --kernel-trace csv (steps are delimited by adam_kernel)."""
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
step = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'adam_kernel' in r['Kernel_Name']]
a, b = idx[step], idx[step + 1]
t0 = int(rows[a]['End_Timestamp'])
for r in rows[a + 1:b + 1]:
    n = re.sub(r'\(.*', '', r['Kernel_Name']).replace('void kws::', '').replace('kws::', '')[:50]
    s = (int(r['Start_Timestamp']) - t0) / 1e3
    e = (int(r['End_Timestamp']) - t0) / 1e3
    print(f"{s:8.1f} {e:8.1f} {e - s:7.1f} q={r['Queue_Id']} vgpr={r['VGPR_Count']:>3} lds={r['LDS_Block_Size']:>6} grid={r['Grid_Size_X']:>7} {n}")
