import subprocess,os,re,sys
# scratch files go under tools/tmp/ (git-ignored), never the cwd
TMP=os.path.join(os.path.dirname(os.path.abspath(__file__)),"tmp"); os.makedirs(TMP,exist_ok=True); os.chdir(TMP)
LLVM="/opt/rocm/lib/llvm/bin"; LIB="/root/repo/sela_amd/libsela_hip.so"
subprocess.check_call([LLVM+"/llvm-objcopy","--dump-section",".hip_fatbin=fat.bin",LIB])
blob=open("fat.bin","rb").read(); magic=b"__CLANG_OFFLOAD_BUNDLE__"; starts=[];at=0
while (at:=blob.find(magic,at))>=0: starts.append(at); at+=1
for k,b in enumerate(starts):
    open(f"b{k}.bin","wb").write(blob[b: starts[k+1] if k+1<len(starts) else len(blob)])
    subprocess.check_call([LLVM+"/clang-offload-bundler","--unbundle","--type=o",f"--input=b{k}.bin","--targets=hipv4-amdgcn-amd-amdhsa--gfx950",f"--output=dev{k}.co"])
    out=subprocess.check_output([LLVM+"/llvm-readelf","--notes",f"dev{k}.co"],text=True)
    cur={}
    for line in out.splitlines():
        l=line.strip().lstrip("- ")
        for key in (".name",".private_segment_fixed_size",".vgpr_count",".vgpr_spill_count",".group_segment_fixed_size",".sgpr_spill_count"):
            if l.startswith(key+":"):
                cur[key]=l.split(":",1)[1].strip()
        if l.startswith(".wavefront_size") :
            n=subprocess.check_output(["c++filt",cur[".name"]],text=True).strip().split("(")[0]
            print(f"{n:50s} vgpr {cur['.vgpr_count']:>4} vspill {cur['.vgpr_spill_count']:>3} sspill {cur['.sgpr_spill_count']:>3} scratch {cur['.private_segment_fixed_size']:>4} lds {cur['.group_segment_fixed_size']}"); cur={}
