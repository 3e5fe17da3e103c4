// membench.hip - ablation of the dense sweep's memory pattern on MI355X (development tool, not product).
// Streams N voxels: reads 4 B (f32) + 1 B (label), writes 1 B, with selectable extra work.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f4v __attribute__((ext_vector_type(4)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n",hipGetErrorString(e),__LINE__); exit(1);} }while(0)

__device__ __forceinline__ double wave_sum(double v){ for(int o=32;o>0;o>>=1) v+=__shfl_xor(v,o,64); return v; }

// MODE 0: copy labels only (1B in,1B out); 1: read I + labels, write labels, no stats (keep alive);
// 2: + integer counts; 3: + f32 sums; 4: + f64 sums (as product); 5: read-only I+labels with f64 sums (no store)
template<int MODE>
__global__ void __launch_bounds__(256) k(const uint8_t* __restrict__ in, uint8_t* __restrict__ out, const float* __restrict__ I,
                                          uint32_t total, double* res) {
    const uint32_t nfull = total >> 10;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t wave = (blockIdx.x*blockDim.x+threadIdx.x)>>6, nw=(gridDim.x*blockDim.x)>>6;
    long long nin=0,nout=0; double s1=0,s2=0; float t1=0,t2=0;
    for (uint32_t u=wave; u<nfull; u+=nw) {
        const uint32_t base=(u<<10)+(lane<<2);
        uint32_t w[4]; f4v f[4];
#pragma unroll
        for(int j=0;j<4;j++){ w[j]=*(const uint32_t*)(in+base+(j<<8)); if(MODE>=1) f[j]=*(const f4v*)(I+base+(j<<8)); }
#pragma unroll
        for(int j=0;j<4;j++){
            uint32_t v=w[j];
            if(MODE!=5) *(uint32_t*)(out+base+(j<<8))=v;
            if(MODE==1){ asm volatile("" :: "v"(f[j][0]),"v"(f[j][1]),"v"(f[j][2]),"v"(f[j][3])); }
            if(MODE>=2){
                uint32_t sb=v&0x01010101u, ob=~(v|(v>>2)|(v>>5))&0x01010101u;
                nin+=__popc(sb); nout+=__popc(ob);
                if(MODE==2){ asm volatile("" :: "v"(f[j][0]),"v"(f[j][1]),"v"(f[j][2]),"v"(f[j][3])); }
                if(MODE==3){
#pragma unroll
                    for(int b=0;b<4;b++){ float x=f[j][b]; t1+=((sb>>(8*b))&1u)?x:0.f; t2+=((ob>>(8*b))&1u)?x:0.f; }
                }
                if(MODE>=4){
#pragma unroll
                    for(int b=0;b<4;b++){ double x=(double)f[j][b]; s1+=((sb>>(8*b))&1u)?x:0.0; s2+=((ob>>(8*b))&1u)?x:0.0; }
                }
            }
        }
    }
    s1+=t1; s2+=t2; s1+= (double)nin; s2+=(double)nout;
    s1=wave_sum(s1); s2=wave_sum(s2);
    if(lane==0 && (s1+s2)==-12345.0) res[0]=s1+s2;
}
// float4 copy reference
__global__ void __launch_bounds__(256) kcopy(const f4v* __restrict__ a, f4v* __restrict__ b, size_t n){
    for(size_t i=(size_t)blockIdx.x*blockDim.x+threadIdx.x;i<n;i+=(size_t)gridDim.x*blockDim.x) b[i]=a[i];
}
template<int MODE> float run(const uint8_t* in,uint8_t* out,const float* I,uint32_t total,double* res,int blocks,int reps){
    hipEvent_t a,b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    k<MODE><<<blocks,256>>>(in,out,I,total,res); CK(hipDeviceSynchronize());
    CK(hipEventRecord(a)); for(int r=0;r<reps;r++) k<MODE><<<blocks,256>>>(in,out,I,total,res); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms,a,b)); return ms/reps;
}
int main(int argc,char**argv){
    uint32_t total = 640u*884u*896u; int reps=20;
    uint8_t *in,*out; float* I; double* res;
    CK(hipMalloc(&in,total)); CK(hipMalloc(&out,total)); CK(hipMalloc(&I,(size_t)total*4)); CK(hipMalloc(&res,8));
    std::vector<uint8_t> h(total); for(uint32_t i=0;i<total;i++) h[i]=(i*2654435761u>>28)==0?1:((i>>10)&1?4:0);
    CK(hipMemcpy(in,h.data(),total,hipMemcpyHostToDevice));
    std::vector<float> hf(1<<20); for(size_t i=0;i<hf.size();i++) hf[i]=(float)(rand()%255)/255.f;
    for(size_t o=0;o<total;o+=hf.size()) CK(hipMemcpy(I+o,hf.data(),std::min<size_t>(hf.size(),total-o)*4,hipMemcpyHostToDevice));
    const char* names[6]={"labels copy only (2B/vox)","I+lab read, lab write, no math (6B/vox)","+ int counts","+ f32 sums","+ f64 sums (product)","read-only I+lab, f64 sums (5B/vox)"};
    double bytes[6]={2,6,6,6,6,5};
    for(int blocks: {1024,2048,4096,8192}){
        printf("blocks %d\n",blocks);
        float ms[6];
        ms[0]=run<0>(in,out,I,total,res,blocks,reps); ms[1]=run<1>(in,out,I,total,res,blocks,reps); ms[2]=run<2>(in,out,I,total,res,blocks,reps);
        ms[3]=run<3>(in,out,I,total,res,blocks,reps); ms[4]=run<4>(in,out,I,total,res,blocks,reps); ms[5]=run<5>(in,out,I,total,res,blocks,reps);
        for(int m=0;m<6;m++) printf("  mode %d %-45s %.4f ms  %.0f GB/s\n",m,names[m],ms[m],bytes[m]*total/ms[m]/1e6);
    }
    {   // float4 copy of 2 GB -> 2 GB
        size_t n=(size_t)total*4/16/2; f4v* a=(f4v*)I; f4v* b=a+n;
        hipEvent_t e0,e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        kcopy<<<4096,256>>>(a,b,n); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); for(int r=0;r<reps;r++) kcopy<<<4096,256>>>(a,b,n); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms,e0,e1)); ms/=reps;
        printf("float4 copy %.1f MB read + write: %.4f ms  %.0f GB/s\n", n*16/1e6, ms, 2.0*n*16/ms/1e6);
    }
    return 0;
}
