"""Per basic-block census of a kernel in an assembly listing (hipcc -S --cuda-device-only): MFMA / VALU / scratch /
barrier counts between labels -- a quick way to see whether spills or stray instructions sit inside the time loop.
usage: python tools/asm_regions.py file.s <substring of the kernel's mangled name>"""
import re
import sys


def main(path, pat):
    s = open(path).read()
    names = [m.group(1) for m in re.finditer(r'^(\S*%s\S*):' % re.escape(pat), s, re.M)]
    for name in names:
        i = s.index(name + ':')
        j = s.index('.Lfunc_end', i)
        body = s[i:j].split('\n')
        print(name, len(body), 'lines')
        marks = [n for n, l in enumerate(body) if re.match(r'^\.LBB\d+_\d+:', l)] + [len(body)]
        for a, b in zip(marks, marks[1:]):
            seg = body[a:b]
            cnt = lambda rx: sum(bool(re.search(rx, l)) for l in seg)
            mf, sc, bar = cnt(r'v_mfma'), cnt(r'scratch_'), cnt(r's_barrier')
            if mf or sc or bar:
                br = [l.split()[-1] for l in seg if re.match(r'\s+s_c?branch', l)]
                print('  %-12s len %4d mfma %3d valu %4d trans %3d ds %3d scratch %3d barrier %d -> %s' % (
                    body[a].rstrip(':'), b - a, mf, cnt(r'^\s+v_') - mf, cnt(r'v_(exp|log|rcp)_f32'), cnt(r'^\s+ds_'), sc, bar, ','.join(br[:3])))


if __name__ == '__main__':
    main(sys.argv[1], sys.argv[2])
