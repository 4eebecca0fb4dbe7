"""Turn a tools/pmc_sweep.sh result (gpurun_out/pmcs_<tag>.txt) into an issue-cycle budget per SIMD:
python tools/pmc_account.py gpurun_out/pmcs_<tag>.txt ["title"]   (units: see profiles/r02_pmc_issue_accounting.txt)"""
import sys
v = {}
for line in open(sys.argv[1]):
    f = line.split()
    if len(f) >= 3:
        try:
            v[f[1]] = float(f[2])
        except ValueError:
            pass
title = sys.argv[2] if len(sys.argv) > 2 else sys.argv[1]
kc = v['GRBM_GUI_ACTIVE'] / 8
simd = kc * 1024
print(title)
print('  kernel cycles (GRBM_GUI_ACTIVE / 8): %.0f k; SIMD-cycles available: %.3g' % (kc / 1e3, simd))
rows = [('MFMA', v['SQ_VALU_MFMA_BUSY_CYCLES']), ('VALU', 4 * v['SQ_ACTIVE_INST_VALU'] ),
        ('LDS issue', 4 * v['SQ_ACTIVE_INST_LDS']), ('scalar', 4 * v['SQ_ACTIVE_INST_SCA']),
        ('VMEM issue', 4 * v['SQ_ACTIVE_INST_VMEM']), ('misc (barrier, waitcnt, ...)', 4 * v['SQ_ACTIVE_INST_MISC'])]
tot = 0.0
for name, c in rows:
    print('    %-30s %5.1f %% of the SIMD cycles' % (name, 100 * c / simd))
    tot += c
print('    %-30s %5.1f %%' % ('sum (issue-busy)', 100 * tot / simd))
m = v['SQ_INSTS_MFMA']
if m <= 0:      # a kernel without matrix instructions (the DSNT head): per-wave-instruction counts instead
    print('  instructions: VALU %.4g, LDS %.4g, scalar %.4g, VMEM %.4g (SQ_INSTS_*; no MFMA)' % (v['SQ_INSTS_VALU'], v['SQ_INSTS_LDS'], v['SQ_INSTS_SALU'], v['SQ_INSTS_VMEM']))
    m = float('nan')
else:
  print('  instructions per MFMA: VALU %.2f, LDS %.2f, scalar %.2f, VMEM %.2f   (SQ_INSTS_*; %.3g MFMAs)' % (
    v['SQ_INSTS_VALU'] / m, v['SQ_INSTS_LDS'] / m, v['SQ_INSTS_SALU'] / m, v['SQ_INSTS_VMEM'] / m, m))
if v['SQ_VALU_MFMA_BUSY_CYCLES'] > 0:
    print('  MFMA and VALU co-executing: %.1f %% of the matrix-pipe time' % (100 * v['SQ_VALU_MFMA_COEXEC_CYCLES'] / v['SQ_VALU_MFMA_BUSY_CYCLES']))
print('  LDS unit: active %.1f %% of the CU cycles, bank conflicts %.1f %% of that, unaligned stalls %.0f' % (
    100 * v['SQ_LDS_IDX_ACTIVE'] / (kc * 256), 100 * v['SQ_LDS_BANK_CONFLICT'] / max(1.0, v['SQ_LDS_IDX_ACTIVE']), v['SQ_LDS_UNALIGNED_STALL']))
print('  waves: waiting (s_waitcnt / barrier) %.0f %% of their resident cycles, issue-stalled %.0f %%, issuing %.0f %%' % (
    100 * v['SQ_WAIT_ANY'] / v['SQ_WAVE_CYCLES'], 100 * v['SQ_WAIT_INST_ANY'] / v['SQ_WAVE_CYCLES'],
    100 * v['SQ_ACTIVE_INST_ANY'] / v['SQ_WAVE_CYCLES']))
