"""Condensed view of a kernel ISA (hipcc --cuda-device-only -S): runs of LDS-DMA / buffer loads / s_waitcnt / barriers / MFMAs / LDS
reads with their line numbers — where does the compiler wait, and for what.  python tools/isa_summary.py kernel.s"""
import sys,re
out=[];prev=None;cnt=0;first=None
for n,l in enumerate(open(sys.argv[1]),1):
    l=l.strip()
    m=re.match(r'(buffer_load_dwordx4|buffer_load_dword\w*|s_waitcnt|s_barrier|v_mfma\w+|ds_read_b128|ds_read_b32|ds_write\w+|buffer_store\w+|s_cbranch\w+|s_sleep|global_load\w+|s_load\w+)\b(.*)',l)
    if not m: continue
    op,rest=m.group(1),m.group(2)
    key=op
    if op=='s_waitcnt': key=op+' '+rest.split(';')[0].strip()
    if op.startswith('buffer_load') and ' lds' in rest: key='DMA->lds'
    if op.startswith('v_mfma'): key='mfma'
    if op.startswith('s_cbranch'): key=op+' '+rest.strip()
    if key==prev: cnt+=1
    else:
        if prev: out.append(f"{first}: {prev} x{cnt}")
        prev=key;cnt=1;first=n
out.append(f"{first}: {prev} x{cnt}")
print("\n".join(out))
