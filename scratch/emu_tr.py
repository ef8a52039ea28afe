import numpy as np
def tile_off(row,ch): return 256*row+16*(ch ^ (((row&3)<<2)|((row>>2)&3)))
V=np.arange(64*128).reshape(64,128)  # value = key*128+d
lds=np.zeros(64*256//2,dtype=np.int64)
for row in range(64):
    for ch in range(16):
        o=tile_off(row,ch)//2
        lds[o:o+8]=V[row,ch*8:ch*8+8]
bad=0
for kk in range(4):
  for d in range(4):
    for half in range(2):
      M={}
      for lane in range(64):
        h=lane>>5; q4=(lane>>2)&3; p4=lane&3; g1=(lane>>4)&1
        v_base=256*(4*h+8*half+q4)+8*(p4&1); v_low=(2*g1+(p4>>1))^(h+2*half)
        addr=v_base+4096*kk+64*(d^q4)+16*v_low
        M[lane]=lds[addr//2:addr//2+4]
      for lane in range(64):
        g=lane>>4; i=lane&15; h=lane>>5; r=lane&31
        out=[M[16*g+4*e+(i>>2)][i&3] for e in range(4)]
        for e in range(4):
          key=16*kk+4*h+8*half+e; dd=32*d+r
          if out[e]!=V[key,dd]:
            bad+=1
            if bad<10: print("mismatch kk",kk,"d",d,"half",half,"lane",lane,"e",e,"got",divmod(out[e],128),"want",(key,dd))
print("bad",bad)
