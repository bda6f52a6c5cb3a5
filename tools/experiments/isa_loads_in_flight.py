import struct,subprocess,sys,re
def analyze(path):
    d=open(path,'rb').read()
    i=0
    while True:
        i=d.find(b'\x7fELF',i)
        if i<0: break
        if i>0 and struct.unpack_from('<H',d,i+18)[0]==224:
            shoff=struct.unpack_from('<Q',d,i+40)[0]; shentsize,shnum=struct.unpack_from('<HH',d,i+58)
            open('/tmp/p0.elf','wb').write(d[i:i+shoff+shentsize*shnum])
            dis=subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-objdump','-d','/tmp/p0.elf'],capture_output=True,text=True).stdout
            out=[]; dist=[]; pend=[]; n=0; stall_events=0
            for line in dis.splitlines():
                if not line.startswith('\t'): 
                    pend=[]; continue
                n+=1
                if 'global_load' in line or 'scratch_load' in line: pend.append(n)
                elif 'global_store' in line or 'scratch_store' in line: pend.append(None)
                m=re.search(r's_waitcnt.*vmcnt\((\d+)\)',line)
                if m:
                    k=int(m.group(1))
                    if len(pend)>k:
                        done=pend[:len(pend)-k]; pend=pend[len(pend)-k:]
                        ld=[x for x in done if x is not None]
                        if ld:
                            stall_events+=1
                            out.append(len(ld)); dist.append(n-max(ld))   # distance from the YOUNGEST load waited for
            print(path.split('_')[-1], 'stall points',stall_events,'loads per stall point %.2f'%(sum(out)/max(1,len(out))),'median distance youngest-load->wait',sorted(dist)[len(dist)//2], 'mean %.1f'%(sum(dist)/len(dist)))
            return
        i+=4
for p in sys.argv[1:]: analyze(p)
