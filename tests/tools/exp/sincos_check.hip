#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include "/root/repo/drl-on-robot-arm_amd/csrc/armenv_kin.h"
__global__ void k(const double* x, double* s, double* c, int n){int i=blockIdx.x*blockDim.x+threadIdx.x; if(i<n) armenv::sincos_joint(x[i], s[i], c[i]);}
int main(){const int n=1<<20; double *x,*s,*c; hipMallocManaged(&x,n*8);hipMallocManaged(&s,n*8);hipMallocManaged(&c,n*8);
 for(int i=0;i<n;i++){ double t=(double)i/n; x[i]= (i%3==0)? (t-0.5)*20.0 : (i%3==1 ? (t-0.5)*2000.0 : (t-0.5)*6.3);} 
 k<<<n/256,256>>>(x,s,c,n); hipDeviceSynchronize(); double ms=0,mc=0; for(int i=0;i<n;i++){ms=fmax(ms,fabs(s[i]-sin(x[i]))); mc=fmax(mc,fabs(c[i]-cos(x[i])));} printf("max err sin %.3e cos %.3e\n",ms,mc); return 0;}
