// Stand-in for <cuda_runtime.h> (test infrastructure, oracle/_ref only).  Lets the reference's host code -- which
// includes CUDA headers and decorates shared functions with __host__ __device__ -- compile with plain g++.  Every
// "device" entry point is a host stub: the GPU-only paths of the reference are never executed by oracle/_ref.
#pragma once
#include <climits>
#include <cmath>
#include <cstddef>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define __host__
#define __device__
#define __global__
#define __shared__ static
#define __constant__ static
#define __forceinline__ inline
#define __align__(n) alignas(n)
#define __launch_bounds__(...)
struct float2 { float x, y; };
struct float3 { float x, y, z; };
struct float4 { float x, y, z, w; };
struct int3 { int x, y, z; };
struct uint2 { unsigned x, y; };
struct uint3 { unsigned x, y, z; };
struct int2 { int x, y; };
struct dim3 {
  unsigned x, y, z;
  dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};
inline float2 make_float2(float x, float y) { return float2{x, y}; }
inline float3 make_float3(float x, float y, float z) { return float3{x, y, z}; }
inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
inline int3 make_int3(int x, int y, int z) { return int3{x, y, z}; }
typedef int cudaError_t;
typedef cudaError_t cudaError;
enum { cudaSuccess = 0, cudaErrorMemoryAllocation = 2 };
enum cudaMemcpyKind { cudaMemcpyHostToHost, cudaMemcpyHostToDevice, cudaMemcpyDeviceToHost, cudaMemcpyDeviceToDevice, cudaMemcpyDefault };
typedef void *cudaStream_t;
typedef void *cudaEvent_t;
#define cudaStreamPerThread ((cudaStream_t)0)
enum { cudaComputeModeDefault = 0, cudaComputeModeProhibited = 2 };
struct cudaDeviceProp { size_t totalGlobalMem; char name[256]; int major, minor, computeMode; };
inline cudaError_t cudaGetLastError() { return cudaSuccess; }
inline cudaError_t cudaPeekAtLastError() { return cudaSuccess; }
inline const char *cudaGetErrorString(cudaError_t) { return "no CUDA in oracle/_ref"; }
inline cudaError_t cudaMalloc(void **p, size_t n) { *p = std::malloc(n); return *p ? cudaSuccess : cudaErrorMemoryAllocation; }
template <class T> inline cudaError_t cudaMalloc(T **p, size_t n) { return cudaMalloc((void **)p, n); }
inline cudaError_t cudaMallocHost(void **p, size_t n) { return cudaMalloc(p, n); }
template <class T> inline cudaError_t cudaMallocHost(T **p, size_t n) { return cudaMalloc((void **)p, n); }
inline cudaError_t cudaFree(void *p) { std::free(p); return cudaSuccess; }
inline cudaError_t cudaFreeHost(void *p) { std::free(p); return cudaSuccess; }
inline cudaError_t cudaMemcpy(void *d, const void *s, size_t n, cudaMemcpyKind) { std::memcpy(d, s, n); return cudaSuccess; }
inline cudaError_t cudaMemcpyAsync(void *d, const void *s, size_t n, cudaMemcpyKind, cudaStream_t = 0) { std::memcpy(d, s, n); return cudaSuccess; }
inline cudaError_t cudaMemset(void *d, int v, size_t n) { std::memset(d, v, n); return cudaSuccess; }
inline cudaError_t cudaMemsetAsync(void *d, int v, size_t n, cudaStream_t = 0) { std::memset(d, v, n); return cudaSuccess; }
inline cudaError_t cudaDeviceSynchronize() { return cudaSuccess; }
inline cudaError_t cudaStreamSynchronize(cudaStream_t) { return cudaSuccess; }
inline cudaError_t cudaSetDevice(int) { return cudaSuccess; }
inline cudaError_t cudaGetDevice(int *d) { *d = 0; return cudaSuccess; }
inline cudaError_t cudaGetDeviceCount(int *d) { *d = 0; return cudaSuccess; }
inline cudaError_t cudaDeviceReset() { return cudaSuccess; }
inline cudaError_t cudaMemGetInfo(size_t *f, size_t *t) { *f = *t = 0; return cudaSuccess; }
inline cudaError_t cudaGetDeviceProperties(cudaDeviceProp *p, int) { std::memset(p, 0, sizeof *p); return cudaSuccess; }
// names that appear inside __global__ / __device__ bodies (parsed, never run)
static const uint3 threadIdx = {0, 0, 0}, blockIdx = {0, 0, 0};
static const dim3 blockDim, gridDim;
inline void __syncthreads() {}
inline void __syncwarp(unsigned = 0xffffffffu) {}
inline void __threadfence() {}
template <class T> inline T __shfl_down_sync(unsigned, T v, int, int = 32) { return v; }
template <class T> inline T __shfl_down(T v, int, int = 32) { return v; }
template <class T> inline T __shfl_sync(unsigned, T v, int, int = 32) { return v; }
template <class T> inline T __shfl_up_sync(unsigned, T v, int, int = 32) { return v; }
template <class T> inline T __ldg(const T *p) { return *p; }
template <class T> inline T atomicAdd(T *a, T v) { T o = *a; *a += v; return o; }
inline float __int_as_float(int v) { float f; std::memcpy(&f, &v, 4); return f; }
inline float rsqrtf(float x) { return 1.0f / std::sqrt(x); }
