// Microbenchmark: what does a dependent chain of small kernels cost on this box?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("ERR %s at %d\n",hipGetErrorString(e),__LINE__); return 1;}}while(0)
__global__ void k_empty(float* p){ if(p && threadIdx.x==1000000) p[0]=1.f; }
__global__ void k_touch(float* p, int n){ int i=blockIdx.x*blockDim.x+threadIdx.x; if(i<n) p[i]+=1.f; }
int main(){
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  float* d; CK(hipMalloc(&d, 1<<24));
  hipEvent_t e0,e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int variant=0; variant<4; ++variant){
    const int N=1000;
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for(int i=0;i<N;++i){
      if(variant==0) k_empty<<<1,64,0,s>>>(d);
      else if(variant==1) k_empty<<<256,256,0,s>>>(d);
      else if(variant==2) k_touch<<<256,256,0,s>>>(d,65536);
      else k_empty<<<2048,256,0,s>>>(d);
    }
    CK(hipStreamEndCapture(s,&g)); CK(hipGraphInstantiate(&ge,g,nullptr,nullptr,0));
    CK(hipGraphLaunch(ge,s)); CK(hipStreamSynchronize(s));
    hipEventRecord(e0,s);
    for(int r=0;r<5;++r) CK(hipGraphLaunch(ge,s));
    hipEventRecord(e1,s); CK(hipStreamSynchronize(s));
    float ms; hipEventElapsedTime(&ms,e0,e1);
    printf("graph variant %d: %.3f us per kernel\n", variant, ms*1e3/(5*N));
    // eager
    hipEventRecord(e0,s);
    for(int i=0;i<N;++i){
      if(variant==0) k_empty<<<1,64,0,s>>>(d);
      else if(variant==1) k_empty<<<256,256,0,s>>>(d);
      else if(variant==2) k_touch<<<256,256,0,s>>>(d,65536);
      else k_empty<<<2048,256,0,s>>>(d);
    }
    hipEventRecord(e1,s); CK(hipStreamSynchronize(s));
    hipEventElapsedTime(&ms,e0,e1);
    printf("eager variant %d: %.3f us per kernel\n", variant, ms*1e3/N);
  }
  return 0;
}
