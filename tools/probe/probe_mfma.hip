// Probe: verify gfx950 MFMA 16x16x32 bf16 lane maps and ds_read_tr16_b64 semantics
// with exact integer data. Diagnostic tool only (not part of the product path).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <cstring>
#include <vector>
typedef __attribute__((ext_vector_type(4))) short s16x4;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf4;

static inline uint16_t f2bf(float f){ uint32_t u; memcpy(&u,&f,4); return (uint16_t)(u>>16);} // exact for small ints

// A [16][32] row-major (k contiguous), B given as Bt [16 cols][32 k] (k contiguous). C[16][16]
__global__ void k_row(const uint16_t* A, const uint16_t* Bt, float* C){
  int l = threadIdx.x;
  bf8 a, b;
  const __bf16* Ap = (const __bf16*)A; const __bf16* Bp=(const __bf16*)Bt;
  for(int j=0;j<8;j++){ a[j]=Ap[(l&15)*32 + 8*(l>>4)+j]; b[j]=Bp[(l&15)*32 + 8*(l>>4)+j]; }
  f32x4 acc={0,0,0,0};
  acc=__builtin_amdgcn_mfma_f32_16x16x32_bf16(a,b,acc,0,0,0);
  for(int r=0;r<4;r++) C[((l>>4)*4+r)*16 + (l&15)] = acc[r];
}
// TN: P [32 m][16 i] (i contiguous), Q [32 m][16 j]; C[i][j] = sum_m P[m][i]Q[m][j] via tr reads
__global__ void k_tr(const uint16_t* P, const uint16_t* Q, float* C, uint16_t* dump){
  __shared__ __attribute__((aligned(16))) uint16_t sP[32*16];
  __shared__ __attribute__((aligned(16))) uint16_t sQ[32*16];
  int l = threadIdx.x;
  for(int i=l;i<512;i+=64){ sP[i]=P[i]; sQ[i]=Q[i]; }
  __syncthreads();
  int g=l>>4, q=(l&15)>>2, p=l&3;
  // lane 4q+p of group g supplies address of row (8g+q), cols 4p..4p+3 ; second read rows 8g+4+q
  typedef s16x4 __attribute__((address_space(3)))* lp;
  s16x4 a0=__builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(sP + (8*g+q)*16 + 4*p));
  s16x4 a1=__builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(sP + (8*g+4+q)*16 + 4*p));
  s16x4 b0=__builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(sQ + (8*g+q)*16 + 4*p));
  s16x4 b1=__builtin_amdgcn_ds_read_tr16_b64_v4i16((lp)(sQ + (8*g+4+q)*16 + 4*p));
  for(int j=0;j<4;j++){ dump[l*8+j]=(uint16_t)a0[j]; dump[l*8+4+j]=(uint16_t)a1[j]; }
  union { bf8 v; s16x4 h[2]; } ua, ub; ua.h[0]=a0; ua.h[1]=a1; ub.h[0]=b0; ub.h[1]=b1;
  f32x4 acc={0,0,0,0};
  acc=__builtin_amdgcn_mfma_f32_16x16x32_bf16(ua.v,ub.v,acc,0,0,0);
  for(int r=0;r<4;r++) C[((l>>4)*4+r)*16 + (l&15)] = acc[r];
}
int main(){
  std::vector<uint16_t> A(16*32), Bt(16*32), P(32*16), Q(32*16);
  std::vector<float> Af(16*32), Bf(16*32), Pf(512), Qf(512);
  srand(1);
  for(int i=0;i<512;i++){ Af[i]=(float)(rand()%7-3); Bf[i]=(float)(rand()%5-2); Pf[i]=(float)(rand()%7-3); Qf[i]=(float)(rand()%5-2);
    A[i]=f2bf(Af[i]); Bt[i]=f2bf(Bf[i]); P[i]=f2bf(Pf[i]); Q[i]=f2bf(Qf[i]); }
  uint16_t *dA,*dB,*dP,*dQ,*dD; float *dC,*dC2;
  hipMalloc(&dA,1024); hipMalloc(&dB,1024); hipMalloc(&dP,1024); hipMalloc(&dQ,1024); hipMalloc(&dD,64*8*2);
  hipMalloc(&dC,1024); hipMalloc(&dC2,1024);
  hipMemcpy(dA,A.data(),1024,hipMemcpyHostToDevice); hipMemcpy(dB,Bt.data(),1024,hipMemcpyHostToDevice);
  hipMemcpy(dP,P.data(),1024,hipMemcpyHostToDevice); hipMemcpy(dQ,Q.data(),1024,hipMemcpyHostToDevice);
  k_row<<<1,64>>>(dA,dB,dC); k_tr<<<1,64>>>(dP,dQ,dC2,dD);
  std::vector<float> C(256),C2(256); std::vector<uint16_t> D(512);
  hipMemcpy(C.data(),dC,1024,hipMemcpyDeviceToHost); hipMemcpy(C2.data(),dC2,1024,hipMemcpyDeviceToHost);
  hipMemcpy(D.data(),dD,1024,hipMemcpyDeviceToHost);
  int bad1=0,bad2=0,bad3=0;
  for(int i=0;i<16;i++)for(int j=0;j<16;j++){ float r=0; for(int k=0;k<32;k++) r+=Af[i*32+k]*Bf[j*32+k]; if(r!=C[i*16+j]) bad1++; }
  for(int i=0;i<16;i++)for(int j=0;j<16;j++){ float r=0; for(int m=0;m<32;m++) r+=Pf[m*16+i]*Qf[m*16+j]; if(r!=C2[i*16+j]) bad2++; }
  // expected dump: lane l elem e (0..7) = P[m=8*(l>>4)+e][i=l&15]
  for(int l=0;l<64;l++)for(int e=0;e<8;e++){ if(D[l*8+e]!=P[(8*(l>>4)+e)*16+(l&15)]) bad3++; }
  printf("PROBE row-mode mismatches=%d  tr-mode mismatches=%d  tr-dump mismatches=%d\n",bad1,bad2,bad3);
  return (bad1||bad2||bad3)?1:0;
}
