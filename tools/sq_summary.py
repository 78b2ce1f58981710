"""Per-kernel SQ counters from rocprofv3 --pmc passes (csv output): per-launch averages of every counter found, plus the derived ratios the north star asks for.

Units (MI355X_MICROARCH.md, "rocprofv3 PMC slots" / cycle-constant notes): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves;
SQ_VALU_MFMA_BUSY_CYCLES and SQ_BUSY_CYCLES count cycles (summed over the SQs that report); SQ_INSTS_* count wave-instructions; SQ_INSTS_VALU_MFMA_MOPS_F64
counts fp64 MFMA operations in units of 512 FLOP.
"""
import csv
import glob
import sys
from collections import defaultdict

sums, counts = defaultdict(lambda: defaultdict(float)), defaultdict(lambda: defaultdict(int))
for dirname in sys.argv[1:]:
    for fn in glob.glob(dirname + '/**/*counter_collection.csv', recursive=True):
        with open(fn) as f:
            for row in csv.DictReader(f):
                name = row['Kernel_Name'].split('(')[0]
                if not name.startswith(('dl_', 'void dl_')): continue
                sums[name][row['Counter_Name']] += float(row['Counter_Value'])
                counts[name][row['Counter_Name']] += 1

for name in sorted(sums):
    avg = {counter: sums[name][counter] / counts[name][counter] for counter in sums[name]}
    print(name[:100])
    print('    launches profiled: %d' % max(counts[name].values()))
    for counter in sorted(avg):
        print('    %-32s %16.1f' % (counter, avg[counter]))

    def ratio(label, num, den, scale=1.):
        if num in avg and den in avg and avg[den] > 0.:
            print('    -> %-44s %10.4f' % (label, scale * avg[num] / avg[den]))

    ratio('MFMA busy / SQ busy cycles (MFMA utilisation)', 'SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_BUSY_CYCLES')
    ratio('LDS bank-conflict cycles / LDS active cycles', 'SQ_LDS_BANK_CONFLICT', 'SQ_LDS_IDX_ACTIVE')
    ratio('issue-stall (WAIT_INST_ANY) / wave cycles', 'SQ_WAIT_INST_ANY', 'SQ_WAVE_CYCLES')
    ratio('parked (WAIT_ANY: waitcnt / barrier) / wave cycles', 'SQ_WAIT_ANY', 'SQ_WAVE_CYCLES')
    ratio('issuing (ACTIVE_INST_ANY) / wave cycles', 'SQ_ACTIVE_INST_ANY', 'SQ_WAVE_CYCLES')
    ratio('VALU issue (ACTIVE_INST_VALU) / wave cycles', 'SQ_ACTIVE_INST_VALU', 'SQ_WAVE_CYCLES')
    ratio('MFMA wave-instructions / VALU wave-instructions', 'SQ_INSTS_MFMA', 'SQ_INSTS_VALU')
    print()
