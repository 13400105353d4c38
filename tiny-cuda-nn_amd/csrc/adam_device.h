// adam_device.h -- one Adam update (optimizers/adam.h:48-119), shared by k_adam (k_misc.hip) and by the gradient kernels that apply
// the update to the chunk they own as they flush it (k_grid_scatter.hip).  One source for both, so that the two routes produce the
// same bits.
#pragma once
#include "tcnn_common.h"

namespace tcnn_amd {

// One parameter, branch-free (selects instead of early returns, so that a wave whose lanes disagree about "skipped" does not
// execute the body twice).  common_debias = debias_table[common_step]; parameters with their own step count look theirs up.
// `updated` reports whether adam.h:76-84 lets this parameter through.
// debias_of(t): the debiasing factor of step t <= common_step (a table lookup: k_adam reads the global table, the gradient kernels a
// window of it they keep in LDS).
template <typename DebiasOf>
__device__ inline void adam_one(const AdamArgs& a, DebiasOf&& debias_of, const float common_debias, const bool is_matrix, const _Float16 g_h, float& w_fp, _Float16& w_h,
                                float& m1, float& m2, uint32_t& step, bool& updated) {
	// loss_scale is a power of two in practice (128): the reciprocal multiply is then exact, i.e. identical to the division
	float gradient = a.inv_loss_scale_exact ? (float)g_h * a.inv_loss_scale : (float)g_h / a.loss_scale;
	updated = is_matrix ? a.optimize_matrix_params != 0 : (a.optimize_non_matrix_params != 0 && gradient != 0);
	const float weight_fp = w_fp;
	if (is_matrix) gradient += a.l2_reg * weight_fp;
	const float gradient_sq = gradient * gradient;
	const float first_moment = a.beta1 * m1 + (1 - a.beta1) * gradient;
	const float second_moment = a.beta2 * m2 + (1 - a.beta2) * gradient_sq;
	float learning_rate = a.learning_rate;
	if (!is_matrix) learning_rate *= a.non_matrix_learning_rate_factor;
	const uint32_t current_step = step + 1;
	float debias = common_debias;
	if (updated && current_step != a.common_step) debias = debias_of(min(current_step, a.common_step)); // a parameter never has more steps than the optimizer
	learning_rate *= debias;
	const float effective_learning_rate = fminf(fmaxf(learning_rate / (sqrtf(second_moment) + a.epsilon), a.lower_lr_bound), a.upper_lr_bound);
	// weight_decay(rel * lr, abs * lr, w), common_device.h:870-873
	const float decayed_weight = (1 - a.relative_weight_decay * learning_rate) * weight_fp - copysignf(a.absolute_weight_decay * learning_rate, weight_fp);
	float new_weight = decayed_weight - effective_learning_rate * first_moment;
	if (a.weight_clipping_magnitude != 0.0f) new_weight = fminf(fmaxf(new_weight, -a.weight_clipping_magnitude), a.weight_clipping_magnitude);
	w_fp = updated ? new_weight : weight_fp;
	w_h = (_Float16)new_weight; // stored only if updated
	m1 = updated ? first_moment : m1;
	m2 = updated ? second_moment : m2;
	step = updated ? current_step : step;
}

} // namespace tcnn_amd
