import ctypes, os, torch, time
here=os.path.dirname(os.path.abspath(__file__))
lib=ctypes.CDLL(os.path.join(here,'libprobe.so'))
maps=[l.split()[-1] for l in open('/proc/self/maps') if 'libamdhip64' in l]
print('hip runtimes mapped:', sorted(set(maps)))
print(torch.cuda.get_device_name(0), torch.cuda.get_device_properties(0))
dev='cuda'
A=torch.randn(32,2,device=dev); B=torch.randn(2,32,device=dev); O=torch.zeros(32,32,device=dev)
st=torch.cuda.current_stream().cuda_stream
rc=lib.probe_mfma(ctypes.c_void_p(O.data_ptr()),ctypes.c_void_p(A.data_ptr()),ctypes.c_void_p(B.data_ptr()),ctypes.c_void_p(st))
torch.cuda.synchronize()
ref=(A.double()@B.double()).float()
print('rc',rc,'mfma maxerr',(O-ref).abs().max().item())
# exactness vs fma chain
ref2=torch.addcmul(A[:,0:1]*B[0:1,:]*0, A[:,0:1], B[0:1,:])
n=1<<30
a=torch.empty(n,dtype=torch.uint8,device=dev).random_(); o=torch.empty_like(a)
lib.probe_copy.argtypes=[ctypes.c_void_p,ctypes.c_void_p,ctypes.c_size_t,ctypes.c_void_p]
for _ in range(3): lib.probe_copy(o.data_ptr(),a.data_ptr(),n,st)
torch.cuda.synchronize()
e0=torch.cuda.Event(enable_timing=True);e1=torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): lib.probe_copy(o.data_ptr(),a.data_ptr(),n,st)
e1.record(); torch.cuda.synchronize()
ms=e0.elapsed_time(e1)/10
print('copy 1GiB: %.3f ms -> %.2f TB/s (r+w)'%(ms, 2*n/ms/1e9))
print('equal', torch.equal(a,o))
# torch conv baseline (MIOpen) quick timing for context
x=torch.randn(256,64,60,60,device=dev); w=torch.randn(64,64,3,3,device=dev)
for _ in range(3): y=torch.nn.functional.conv2d(x,w,padding=1)
torch.cuda.synchronize(); e0.record()
for _ in range(10): y=torch.nn.functional.conv2d(x,w,padding=1)
e1.record(); torch.cuda.synchronize(); ms=e0.elapsed_time(e1)/10
print('MIOpen fp32 conv3x3 256x64x60x60: %.3f ms -> %.1f TF'%(ms, 256*64*64*9*3600*2/ms/1e9))
os.system('rocminfo | grep -i -E "Compute Unit|Max Clock|gfx|LDS|Wavefront" | head -20; nproc; free -g | head -2; lscpu | grep -i "model name"')
