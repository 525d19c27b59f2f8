import ctypes as C, struct, sys
sys.path.insert(0,'.')
import mcpar_amd as M
L=M.load()
def bits(x): return struct.unpack('<I',struct.pack('<f',x))[0]
nb,first=C.c_uint64(0),C.c_uint32(0)
for e in range(-126,21,6):
    lo,hi=bits(2.0**e),bits(2.0**min(e+6,20))
    L.mcx_debug_sqrt_sweep(lo,hi,C.byref(nb),C.byref(first))
    print(e,nb.value,hex(first.value))
