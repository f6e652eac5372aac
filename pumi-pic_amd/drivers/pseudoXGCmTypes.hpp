// pseudoXGCmTypes.hpp -- particle types of the pseudoXGCm driver (test/pseudoXGCmTypes.hpp:1-33), on the
// MI355X-native particle_structs mirror.
#pragma once
#include "../include/pumipic_adjacency.hpp"

using particle_structs::lid_t;
using particle_structs::MemberTypes;
using particle_structs::SellCSigma;
using pumipic::fp_t;
using pumipic::Vector3d;

// gyro-ring points: start position, end position, id (pseudoXGCmTypes.hpp:18)
typedef MemberTypes<Vector3d, Vector3d, int> Point;
typedef ps::ParticleStructure<Point> PSpt;
// particles: position now / after the push, particle id, ellipse semi-axis b, ellipse angle phi (:28)
typedef MemberTypes<Vector3d, Vector3d, int, float, float> Particle;
typedef ps::ParticleStructure<Particle> PS;
