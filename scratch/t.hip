#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
    unsigned lane = threadIdx.x;
    // push: lanes 3,10,20 push their lane id to lanes 0,1,2; others push 0 to lane 63
    bool is = lane==3||lane==10||lane==20;
    unsigned rank = lane==3?0:lane==10?1:2;
    unsigned dst = is ? rank : 63u;
    unsigned a = (unsigned)__builtin_amdgcn_ds_permute((int)(dst*4), (int)(is ? lane+100 : 0u));
    unsigned b = (unsigned)__builtin_amdgcn_ds_bpermute((int)(((lane+5)&63)*4), (int)(lane*2));
    unsigned c = (unsigned)__builtin_amdgcn_update_dpp((int)777, (int)lane, 0x138, 0xf, 0xf, false);
    out[lane] = a; out[64+lane] = b; out[128+lane]=c;
}
int main(){ unsigned* d; hipMalloc(&d, 192*4); k<<<1,64>>>(d); unsigned h[192]; hipMemcpy(h,d,sizeof h,hipMemcpyDeviceToHost);
 printf("permute: "); for(int i=0;i<6;i++) printf("%u ",h[i]); printf("... l63=%u\n",h[63]);
 printf("bpermute: "); for(int i=0;i<4;i++) printf("%u ",h[64+i]); printf("l62=%u\n", h[64+62]);
 printf("wave_shr1: "); for(int i=0;i<4;i++) printf("%u ",h[128+i]); printf("l16=%u l32=%u\n",h[128+16],h[128+32]); return 0; }
