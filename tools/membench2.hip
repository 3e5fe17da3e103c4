// membench2.hip - read-only variants of the dense recount's access shape (development tool).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f4v __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__); exit(1);} }while(0)
__device__ __forceinline__ double wave_sum(double v){ for(int o=32;o>0;o>>=1) v+=__shfl_xor(v,o,64); return v; }
__device__ __forceinline__ void stats(long long& nin,long long& nout,double& s1,double& s2,uint32_t v,f4v f){
    uint32_t sb=v&0x01010101u, ob=~(v|(v>>2)|(v>>5))&0x01010101u; nin+=__popc(sb); nout+=__popc(ob);
#pragma unroll
    for(int b=0;b<4;b++){ double x=(double)f[b]; s1+=((sb>>(8*b))&1u)?x:0.0; s2+=((ob>>(8*b))&1u)?x:0.0; }
}
// UNITS: 1-KiB units per trip per wave; NT: nontemporal loads
template<int UNITS,bool NT>
__global__ void __launch_bounds__(256) k(const uint8_t* __restrict__ in,const float* __restrict__ I,uint32_t total,double* res){
    const uint32_t nfull=total>>10, lane=threadIdx.x&63;
    const uint32_t wave=(blockIdx.x*blockDim.x+threadIdx.x)>>6, nw=(gridDim.x*blockDim.x)>>6;
    long long nin=0,nout=0; double s1=0,s2=0;
    for(uint32_t u=wave*UNITS; u+UNITS<=nfull; u+=nw*UNITS){
        uint32_t w[4*UNITS]; f4v f[4*UNITS];
#pragma unroll
        for(int q=0;q<UNITS;q++){
            const uint32_t base=((u+q)<<10)+(lane<<2);
#pragma unroll
            for(int j=0;j<4;j++){
                if(NT){ w[4*q+j]=__builtin_nontemporal_load((const uint32_t*)(in+base+(j<<8))); f[4*q+j]=__builtin_nontemporal_load((const f4v*)(I+base+(j<<8))); }
                else { w[4*q+j]=*(const uint32_t*)(in+base+(j<<8)); f[4*q+j]=*(const f4v*)(I+base+(j<<8)); }
            }
        }
#pragma unroll
        for(int j=0;j<4*UNITS;j++) stats(nin,nout,s1,s2,w[j],f[j]);
    }
    s1+=(double)nin; s2+=(double)nout; s1=wave_sum(s1); s2=wave_sum(s2);
    if(lane==0 && (s1+s2)==-12345.0) res[0]=s1+s2;
}
template<int UNITS,bool NT> float run(const uint8_t* in,const float* I,uint32_t total,double* res,int blocks,int reps){
    hipEvent_t a,b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    k<UNITS,NT><<<blocks,256>>>(in,I,total,res); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); for(int r=0;r<reps;r++) k<UNITS,NT><<<blocks,256>>>(in,I,total,res); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms,a,b)); return ms/reps;
}
int main(){
    uint32_t total=640u*884u*896u; int reps=20;
    uint8_t* in; float* I; double* res;
    CK(hipMalloc(&in,total)); CK(hipMalloc(&I,(size_t)total*4)); CK(hipMalloc(&res,8));
    CK(hipMemset(in,0,total)); CK(hipMemset(I,0,(size_t)total*4));
    for(int blocks: {256,512,1024,2048}){
        printf("blocks %4d:  u1 %.4f  u1nt %.4f  u2 %.4f  u2nt %.4f  u4 %.4f ms   (5 B/vox: %.0f GB/s best)\n",blocks,
            run<1,false>(in,I,total,res,blocks,reps),run<1,true>(in,I,total,res,blocks,reps),run<2,false>(in,I,total,res,blocks,reps),
            run<2,true>(in,I,total,res,blocks,reps),run<4,false>(in,I,total,res,blocks,reps), 5.0*total/0.42/1e6);
    }
    return 0;
}
