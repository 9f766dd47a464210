/* Exhaustive check behind tests/test_oracle_extract.py::test_sincosf_restatement_equals_host_libm: the restatement of glibc sinf / cosf
   (the algorithm the GPU describe kernel runs) against the host libm on EVERY float of [0, 6.5).
   gcc -O2 -ffp-contract=off tools/sincosf_exhaustive.c -lm && ./a.out   (also with -mfma -ffp-contract=fast): 0 mismatches. */
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
typedef struct { double sign[4]; double hpi_inv, hpi, c0,c1,c2,c3,c4, s1,s2,s3; } sincos_t;
static const sincos_t T[2] = {
 {{1.0,-1.0,-1.0,1.0}, 0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0, 0x1p0, -0x1.ffffffd0c621cp-2, 0x1.55553e1068f19p-5, -0x1.6c087e89a359dp-10, 0x1.99343027bf8c3p-16, -0x1.555545995a603p-3, 0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13},
 {{1.0,-1.0,-1.0,1.0}, 0x1.45F306DC9C883p+23, 0x1.921FB54442D18p0, -0x1p0, 0x1.ffffffd0c621cp-2, -0x1.55553e1068f19p-5, 0x1.6c087e89a359dp-10, -0x1.99343027bf8c3p-16, -0x1.555545995a603p-3, 0x1.1107605230bc4p-7, -0x1.994eb3774cf24p-13}};
static inline uint32_t abstop12(float x){ uint32_t u; memcpy(&u,&x,4); return (u>>20)&0x7ff; }
static inline float poly(double x, double x2, const sincos_t* p, int n){
  if ((n&1)==0){ double x3=x*x2; double s1=p->s2+x2*p->s3; double x7=x3*x2; double s=x+x3*p->s1; return s+x7*s1; }
  else { double x4=x2*x2; double c2=p->c3+x2*p->c4; double c1=p->c0+x2*p->c1; double x6=x4*x2; double c=c1+x4*p->c2; return c+x6*c2; }
}
static inline double reduce_fast(double x, const sincos_t* p, int* np){ double r=x*p->hpi_inv; int n=((int32_t)r+0x800000)>>24; *np=n; return x-n*p->hpi; }
float re_cosf(float y){ double x=y; int n; const sincos_t* p=&T[0];
  if (abstop12(y)<abstop12(0x1.921FB6p-1f)){ double x2=x*x; if (abstop12(y)<abstop12(0x1p-12f)) return 1.0f; return poly(x,x2,p,1);} 
  x=reduce_fast(x,p,&n); double s=p->sign[n&3]; if (n&2) p=&T[1]; return poly(x*s,x*x,p,n^1); }
float re_sinf(float y){ double x=y; int n; const sincos_t* p=&T[0];
  if (abstop12(y)<abstop12(0x1.921FB6p-1f)){ double s=x*x; if (abstop12(y)<abstop12(0x1p-12f)) return y; return poly(x,s,p,0);} 
  x=reduce_fast(x,p,&n); double s=p->sign[n&3]; if (n&2) p=&T[1]; return poly(x*s,x*x,p,n); }
int main(){ uint64_t n=0,dc=0,ds=0; float hi=6.5f; uint32_t uhi; memcpy(&uhi,&hi,4);
  for (uint32_t u=0; u<uhi; ++u){ float a; memcpy(&a,&u,4); n++; dc+=(cosf(a)!=re_cosf(a)); ds+=(sinf(a)!=re_sinf(a)); }
  printf("n=%llu cos mismatch=%llu sin mismatch=%llu\n",(unsigned long long)n,(unsigned long long)dc,(unsigned long long)ds); return 0; }
