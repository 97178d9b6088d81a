import csv, glob, sys
f = glob.glob('gpurun_out/prof_c4/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'downconv' in r['Kernel_Name']]
i0 = idx[-9]
base = int(rows[i0]['Start_Timestamp'])
for r in rows[i0 - 2:]:
    if 'smeter' in r['Kernel_Name']: continue
    print("%-42s grid %7s q %3s start %8.1f end %8.1f dur %7.1f" % (r['Kernel_Name'][:42], r['Grid_Size_X'], r.get('Queue_Id', '?'), (int(r['Start_Timestamp']) - base) / 1e3, (int(r['End_Timestamp']) - base) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3))
