#!/usr/bin/env python3
"""Linker version script of libssnode.so from the published header: the functions include/ssnode_mi355x.h declares are
the library's dynamic symbols and nothing else is (kernel stubs, launch templates and the HIP registration objects stay
local; the runtime finds kernels through the registration its static constructors make, not through the symbol table).
usage: gen_export_map.py <header> <out.map>"""
import re
import sys


def declared(header_text):
    text = re.sub(r'/\*.*?\*/', '', header_text, flags=re.S)
    names = re.findall(r'^\s*(?:int|long|size_t|double|const char \*|const char\*)\s*\*?\s*([a-z_0-9]+)\s*\(', text, flags=re.M)
    return sorted(set(names))


if __name__ == '__main__':
    names = declared(open(sys.argv[1]).read())
    with open(sys.argv[2], 'w') as f:
        f.write('{\n  global:\n')
        for n in names:
            f.write('    %s;\n' % n)
        f.write('  local:\n    *;\n};\n')
