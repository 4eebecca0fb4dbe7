"""Scan the device code of a built library for the packed-fp32 form that loses results on gfx950 (DESIGN.md "round 3",
profiles/r03_slp_packed_add_hazard.txt): a v_pk_{add,mul,fma}_f32 whose destination register pair overlaps a
VECTOR source pair that is read with crossed halves (op_sel = 1 or op_sel_hi = 0 on that operand: the low result reads the
high half or the reverse).  python tools/check_isa_packed_f32.py [libdsnt_hip.so]  -> exit 1 if found."""
import glob, os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = '/opt/rocm/lib/llvm/bin/'


def scan(lib):
    out = []
    with tempfile.TemporaryDirectory() as d:
        tmp = os.path.join(d, 'lib.so')
        os.symlink(os.path.abspath(lib), tmp)
        subprocess.run([LLVM + 'llvm-objdump', '--offloading', tmp], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        pk = re.compile(r'(v_pk_(?:add|mul|fma)_f32) v\[(\d+):(\d+)\], (.*)')
        total = 0
        for co in sorted(glob.glob(tmp + '.*gfx950')):
            dis = subprocess.run([LLVM + 'llvm-objdump', '-d', co], check=True, capture_output=True, text=True).stdout
            func = '?'
            for line in dis.split('\n'):
                m = re.match(r'^[0-9a-f]+ <(.*)>:', line)
                if m:
                    func = m.group(1)
                m = pk.search(line)
                if not m:
                    continue
                total += 1
                if 'op_sel' not in line:
                    continue
                dlo, dhi = int(m.group(2)), int(m.group(3))
                rest = m.group(4)
                ops = [o.strip() for o in re.split(r',\s*(?![^\[]*\])', rest.split(' op_sel')[0])]
                sel = re.search(r'op_sel:\[([0-9,]+)\]', rest)
                selhi = re.search(r'op_sel_hi:\[([0-9,]+)\]', rest)
                sel = [int(v) for v in sel.group(1).split(',')] if sel else [0] * len(ops)
                selhi = [int(v) for v in selhi.group(1).split(',')] if selhi else [1] * len(ops)
                for i, o in enumerate(ops):
                    r = re.match(r'v\[(\d+):(\d+)\]', o)
                    if not r or i >= len(sel) or i >= len(selhi):
                        continue
                    overlaps = int(r.group(1)) <= dhi and int(r.group(2)) >= dlo
                    crossed = sel[i] == 1 or selhi[i] == 0      # low result reads the high half, or the reverse
                    if overlaps and crossed:
                        out.append((func, line.strip().split('//')[0].strip()))
                        break
    return total, out


if __name__ == '__main__':
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, 'dsnt-pose2d_amd', 'csrc', 'libdsnt_hip.so')
    total, bad = scan(lib)
    print('%s: %d packed fp32 instructions, %d with crossed halves on an operand that overlaps the destination' % (lib, total, len(bad)))
    for f, l in bad[:20]:
        print('  ', f[:70], '|', l)
    sys.exit(1 if bad else 0)
