// core_types.h -- ABI-visible value types of the ba_tracker API.
//
// The reference passes these by value / by pointer across its launcher boundary,
// so their memory layout is part of the drop-in contract:
//   VectorX<T,n> {int nDim; T values[n];}      core/common/Vector.h:11-16
//   Vector2d = 24 B, Vector3d = 32 B            core/common/Vector.h:18,72
//   VectorX<double,4> = 40 B, VectorX<int,2> = 12 B
//   Quaterniond {x,y,z,w} = 32 B                core/common/Quaternion.h:13-18
//   FLOAT = double                              core/common/CustomType.h:6
// (sizes checked by static_asserts below and against the reference in tests).
#ifndef MBAVO_CORE_TYPES_H
#define MBAVO_CORE_TYPES_H

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define MBAVO_HD __host__ __device__ inline
#else
#define MBAVO_HD inline
#endif

namespace SLAM
{
    typedef double FLOAT;

    namespace Core
    {
        template <class T, int nDim_>
        struct VectorX
        {
            int nDim;
            T values[nDim_];
        };

        struct Vector2d : public VectorX<double, 2>
        {
            MBAVO_HD Vector2d() { nDim = 2; }
            MBAVO_HD Vector2d(double x, double y) { nDim = 2; values[0] = x; values[1] = y; }
            MBAVO_HD double &operator()(int i) { return values[i]; }
            MBAVO_HD double operator()(int i) const { return values[i]; }
        };

        struct Vector3d : public VectorX<double, 3>
        {
            MBAVO_HD Vector3d() { nDim = 3; }
            MBAVO_HD Vector3d(double x, double y, double z) { nDim = 3; values[0] = x; values[1] = y; values[2] = z; }
            MBAVO_HD double &operator()(int i) { return values[i]; }
            MBAVO_HD double operator()(int i) const { return values[i]; }
        };

        struct Quaterniond
        {
            double x, y, z, w;
            MBAVO_HD Quaterniond() : x(0), y(0), z(0), w(1) {}
            MBAVO_HD Quaterniond(double x_, double y_, double z_, double w_) : x(x_), y(y_), z(z_), w(w_) {}
        };

        static_assert(sizeof(Vector2d) == 24, "Vector2d must keep the reference's 24-byte stride");
        static_assert(sizeof(Vector3d) == 32, "Vector3d layout");
        static_assert(sizeof(VectorX<double, 4>) == 40, "VectorX<double,4> layout");
        static_assert(sizeof(VectorX<int, 2>) == 12, "VectorX<int,2> layout");
        static_assert(sizeof(Quaterniond) == 32, "Quaterniond layout");
    } // namespace Core
} // namespace SLAM

#endif
