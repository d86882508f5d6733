MS=128,192 KN=4096x28672,4096x14336,8192x16384 python tools/sweep_decode.py 0 2 5 14 21 25
MS=768 KN=3072x3072,1024x3072,2048x3072,512x3072 python tools/sweep_decode.py 0 2 5 14 21 25
MS=1024 KN=1024x4096,2048x4096,512x4096,256x4096 python tools/sweep_decode.py 0 2 5 14 21 25
MS=1088,1152,1280,1536 KN=4096x4096,8192x4096 python tools/sweep_decode.py 0 2 5 21 25
MS=520,576 KN=4096x4096,4096x8192 python tools/sweep_decode.py 0 2 5 21 25
OUT=f32 MS=640,1024 KN=4096x4096,8192x4096,14336x4096 python tools/sweep_decode.py 0 2 5 21 25
OUT=f32 MS=512 KN=8192x8192,4096x8192 python tools/sweep_decode.py 0 2 5 21 25
