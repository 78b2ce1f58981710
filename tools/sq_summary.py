"""Per-kernel SQ counters from rocprofv3 --pmc passes (csv output): per-launch averages of every counter found, plus the derived ratios the north star asks for.

Units (MI355X_MICROARCH.md, "rocprofv3 PMC slots" / cycle-constant notes): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves;
SQ_VALU_MFMA_BUSY_CYCLES and SQ_BUSY_CYCLES count cycles (summed over the SQs that report); SQ_INSTS_* count wave-instructions; SQ_INSTS_VALU_MFMA_MOPS_F64
counts fp64 MFMA operations in units of 512 FLOP.
MFMA utilisation is normalised by the kernel duration of a --kernel-trace --stats pass of the same command (--stats <csv>): busy cycles / (1024 SIMDs x cycles).
"""
import csv
import glob
import sys
from collections import defaultdict

CLOCK_HZ, N_SIMD = 2.4e9, 1024        # MI355X: 256 CUs x 4 SIMDs at the 2.4 GHz peak engine clock (78.6 TFLOP/s fp64 = 1024 SIMDs x 32 FLOP/cycle x 2.4 GHz)
durations = {}                         # kernel -> average duration [s] from a rocprofv3 --kernel-trace --stats csv (--stats file)
args = sys.argv[1:]
if '--stats' in args:
    index = args.index('--stats')
    with open(args[index + 1]) as f:
        for row in csv.DictReader(f):
            durations[row['Name'].split('(')[0]] = float(row['AverageNs']) * 1e-9
    args = args[:index] + args[index + 2:]
sums, counts = defaultdict(lambda: defaultdict(float)), defaultdict(lambda: defaultdict(int))
for dirname in args:
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

    if name in durations:
        cycles = durations[name] * CLOCK_HZ
        print('    kernel duration (rocprofv3 --kernel-trace --stats)   %10.2f us = %.0f cycles at 2.4 GHz' % (1e6 * durations[name], cycles))
        if avg.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.) > 0.:
            print('    -> %-44s %10.4f' % ('MFMA utilisation = MFMA busy cycles / (1024 SIMDs x kernel cycles)', avg['SQ_VALU_MFMA_BUSY_CYCLES'] / (N_SIMD * cycles)))
        if avg.get('SQ_INSTS_VALU_MFMA_MOPS_F64', 0.) > 0.:
            print('    -> %-44s %10.2f' % ('fp64 MFMA TFLOP/s (MOPS_F64 x 512 FLOP / duration)', avg['SQ_INSTS_VALU_MFMA_MOPS_F64'] * 512. / durations[name] / 1e12))
        if avg.get('SQ_ACTIVE_INST_VALU', 0.) > 0.:
            print('    -> %-44s %10.4f' % ('VALU issue = 4 x ACTIVE_INST_VALU quad-cycles / (1024 SIMDs x kernel cycles)', 4. * avg['SQ_ACTIVE_INST_VALU'] / (N_SIMD * cycles)))
    ratio('LDS bank-conflict cycles / LDS active cycles', 'SQ_LDS_BANK_CONFLICT', 'SQ_LDS_IDX_ACTIVE')
    ratio('issue-stall (WAIT_INST_ANY) / wave cycles', 'SQ_WAIT_INST_ANY', 'SQ_WAVE_CYCLES')
    ratio('parked (WAIT_ANY: waitcnt / barrier) / wave cycles', 'SQ_WAIT_ANY', 'SQ_WAVE_CYCLES')
    ratio('issuing (ACTIVE_INST_ANY) / wave cycles', 'SQ_ACTIVE_INST_ANY', 'SQ_WAVE_CYCLES')
    ratio('VALU issue (ACTIVE_INST_VALU) / wave cycles', 'SQ_ACTIVE_INST_VALU', 'SQ_WAVE_CYCLES')
    ratio('MFMA wave-instructions / VALU wave-instructions', 'SQ_INSTS_MFMA', 'SQ_INSTS_VALU')
    print()
