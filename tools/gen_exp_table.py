"""Regenerate rl_markets_amd/csrc/lob_exp_table.h from the `__exp_data` object of the local libm.so.6 (glibc 2.35): located by
the bit pattern of its first member, InvLn2N = 0x1.71547652b82fep0 * 128; layout of sysdeps/ieee754/dbl-64/math_config.h
struct exp_data: invln2N, shift, negln2hiN, negln2loN, poly[4], exp2_shift, exp2_poly[5], tab[2 * 128].  Prints the constants."""
import struct
import sys

path = sys.argv[1] if len(sys.argv) > 1 else "/lib/x86_64-linux-gnu/libm.so.6"
b = open(path, "rb").read()
p = b.find(struct.pack("<d", float.fromhex("0x1.71547652b82fep0") * 128))
assert p >= 0, "no __exp_data in " + path
print([v.hex() for v in struct.unpack("<8d", b[p:p + 64])])
tab = struct.unpack("<256Q", b[p + 14 * 8:p + 14 * 8 + 256 * 8])
assert tab[0] == 0 and tab[1] == 0x3ff0000000000000
for i in range(0, 256, 4):
    print("    " + ", ".join("0x%016xull" % v for v in tab[i:i + 4]) + ", \\")
