// host/posbias.h — the host part of --posBias: transcript length classes, the 20-bin read-start models and the cubic
// spline that turns a model into per-position weights.  The device evaluates the splines (hip/bias.hip); everything here is a few
// hundred flops per job.
#pragma once
#include <cstdint>
#include <vector>
struct sq_index;
#define SQ_POS_BINS 20
#define SQ_POS_CLASSES 5
#define SQ_POS_KNOTS 22   // the 20 bins + one knot at each end (SimplePosBias.cpp:60-80)
struct sq_pos_spline { double x[SQ_POS_KNOTS], y[SQ_POS_KNOTS], a[SQ_POS_KNOTS], b[SQ_POS_KNOTS], c[SQ_POS_KNOTS]; };
// Transcript::lengthClassIndex for every reference (ReadExperiment.inl:352-388 over the non-decoy lengths :152); returns the class count
int sq_pos_length_classes(const sq_index* idx, uint32_t* quantiles /*[5]*/, std::vector<uint8_t>& cls);
// SimplePosBias::finalize on linear masses: the normalised masses (may be null) and the spline through the 22 knots
void sq_pos_finalize(const double* mass /*[20]*/, sq_pos_spline* out, double* norm /*[20] or null*/);
