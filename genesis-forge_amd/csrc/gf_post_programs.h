// gf_post_programs.h — static programs of the fused post-physics kernel (see gf_post_ws.h, point 2).
//
// A program pins the STRUCTURE of one task configuration: the opcode sequence of the termination and reward tables with the
// flags / view slots that select code paths, the command widths and the observation layout.  Numbers (weights, thresholds,
// ranges, scales, rows, pointers, sizes, seeds) stay run-time kernel arguments.  program_matches<P>() compares a packed
// descriptor against the signature field by field; only an exact match launches post_ws_kernel<P>, anything else runs the
// table interpreter.  gf_post_physics_describe() prints a packed descriptor in exactly this notation, so registering a new
// task is: run it once, paste the printed struct here, add it to GF_POST_PROGRAMS.
#pragma once

#include "gf_post_ws.h"

namespace gf {

// examples/command_direction (Go2, 12 DOF): genesis-forge's headline locomotion task and BASELINE.json's benchmark config
struct ProgGo2CommandDirection {
    static constexpr bool kStatic = true;
    static constexpr const char* name = "go2_command_direction";
    static constexpr int DV = 3;
    static constexpr int n_term = 2;
    static constexpr TermSig term[n_term] = {{GF_T_TIMEOUT, GF_TERM_FLAG_TIME_OUT}, {GF_T_BAD_ORIENTATION, 0}};
    static constexpr int n_rew = 6;
    static constexpr RewSig rew[n_rew] = {{GF_R_BASE_HEIGHT, 0, 0, 0},       {GF_R_CMD_TRACK_LIN_VEL, 0, 0, 0}, {GF_R_CMD_TRACK_ANG_VEL, 0, 0, 2},
                                          {GF_R_LIN_VEL_Z_L2, 0, 0, 0},      {GF_R_ACTION_RATE_L2, 0, 0, 0},    {GF_R_DOF_SIMILAR_TO_DEFAULT, 0, 0, 0}};
    static constexpr int n_cmd = 1;
    static constexpr int cmd_width[GF_POST_MAX_CMD] = {3, 0};
    static constexpr int n_obs = 1;
    static constexpr int obs_width[GF_POST_MAX_OBS] = {48, 0};
    static constexpr int obs_history[GF_POST_MAX_OBS] = {1, 0};
    static constexpr int obs_items[GF_POST_MAX_OBS] = {7, 0};
    static constexpr ItemSig item[GF_POST_MAX_OBS][kPostMaxItems] = {{{GF_O_COMMAND, 3, 0, false, false},
                                                                      {GF_O_ANG_VEL_BODY, 3, 0, false, false},
                                                                      {GF_O_LIN_VEL_BODY, 3, 0, false, false},
                                                                      {GF_O_PROJ_GRAVITY, 3, 0, false, false},
                                                                      {GF_O_DOF_POS, 12, 0, false, false},
                                                                      {GF_O_DOF_VEL, 12, 0, true, false},
                                                                      {GF_O_ACTIONS, 12, 0, false, false}},
                                                                     {}};
    static constexpr int n_air = 0;
    static constexpr int n_gait = 0;
};


// examples/simple (Go2): static target command passed as tensor views (no command manager), observation scales, O = 45
struct ProgGo2Simple {
    static constexpr bool kStatic = true;
    static constexpr const char* name = "go2_simple";
    static constexpr int DV = 3;
    static constexpr int n_term = 2;
    static constexpr TermSig term[n_term] = {{GF_T_TIMEOUT, GF_TERM_FLAG_TIME_OUT}, {GF_T_BAD_ORIENTATION, 0}};
    static constexpr int n_rew = 6;
    static constexpr RewSig rew[n_rew] = {{GF_R_BASE_HEIGHT, 0, 0, 0}, {GF_R_CMD_TRACK_LIN_VEL, 0, 0, 0}, {GF_R_CMD_TRACK_ANG_VEL, 0, 1, 0},
                                          {GF_R_LIN_VEL_Z_L2, 0, 0, 0}, {GF_R_ACTION_RATE_L2, 0, 0, 0}, {GF_R_DOF_SIMILAR_TO_DEFAULT, 0, 0, 0}};
    static constexpr int n_cmd = 0;
    static constexpr int cmd_width[GF_POST_MAX_CMD] = {0, 0};
    static constexpr int n_obs = 1;
    static constexpr int obs_width[GF_POST_MAX_OBS] = {45, 0};
    static constexpr int obs_history[GF_POST_MAX_OBS] = {1, 0};
    static constexpr int obs_items[GF_POST_MAX_OBS] = {6, 0};
    static constexpr ItemSig item[GF_POST_MAX_OBS][kPostMaxItems] = {{{GF_O_ANG_VEL_BODY, 3, 0, true, false},
                                                                      {GF_O_LIN_VEL_BODY, 3, 0, true, false},
                                                                      {GF_O_PROJ_GRAVITY, 3, 0, false, false},
                                                                      {GF_O_DOF_POS, 12, 0, false, false},
                                                                      {GF_O_DOF_VEL, 12, 0, true, false},
                                                                      {GF_O_ACTIONS, 12, 0, false, false}},
                                                                     {}};
    static constexpr int n_air = 0;
    static constexpr int n_gait = 0;
};

// examples/contacts (Go2): feet_air_time on the calves (one air-time ContactManager), flat orientation
struct ProgGo2Contacts {
    static constexpr bool kStatic = true;
    static constexpr const char* name = "go2_contacts";
    static constexpr int DV = 3;
    static constexpr int n_term = 2;
    static constexpr TermSig term[n_term] = {{GF_T_TIMEOUT, GF_TERM_FLAG_TIME_OUT}, {GF_T_BAD_ORIENTATION, 0}};
    static constexpr int n_rew = 8;
    static constexpr RewSig rew[n_rew] = {{GF_R_FEET_AIR_TIME, 0, 0, 0}, {GF_R_CMD_TRACK_LIN_VEL, 0, 0, 0}, {GF_R_CMD_TRACK_ANG_VEL, 0, 0, 2}, {GF_R_LIN_VEL_Z_L2, 0, 0, 0},
                                          {GF_R_ANG_VEL_XY_L2, 0, 0, 0}, {GF_R_ACTION_RATE_L2, 0, 0, 0}, {GF_R_DOF_SIMILAR_TO_DEFAULT, 0, 0, 0}, {GF_R_FLAT_ORIENTATION_L2, 0, 0, 0}};
    static constexpr int n_cmd = 1;
    static constexpr int cmd_width[GF_POST_MAX_CMD] = {3, 0};
    static constexpr int n_obs = 1;
    static constexpr int obs_width[GF_POST_MAX_OBS] = {48, 0};
    static constexpr int obs_history[GF_POST_MAX_OBS] = {1, 0};
    static constexpr int obs_items[GF_POST_MAX_OBS] = {7, 0};
    static constexpr ItemSig item[GF_POST_MAX_OBS][kPostMaxItems] = {{{GF_O_COMMAND, 3, 0, false, false},
                                                                      {GF_O_ANG_VEL_BODY, 3, 0, false, false},
                                                                      {GF_O_LIN_VEL_BODY, 3, 0, false, false},
                                                                      {GF_O_PROJ_GRAVITY, 3, 0, false, false},
                                                                      {GF_O_DOF_POS, 12, 0, false, false},
                                                                      {GF_O_DOF_VEL, 12, 0, true, false},
                                                                      {GF_O_ACTIONS, 12, 0, false, false}},
                                                                     {}};
    static constexpr int n_air = 1;
    static constexpr int n_gait = 0;
};

// examples/rough_terrain (Go2, BASELINE config 3): out_of_bounds, undesired contacts, terminated penalty, terrain spawn on reset
struct ProgGo2RoughTerrain {
    static constexpr bool kStatic = true;
    static constexpr const char* name = "go2_rough_terrain";
    static constexpr int DV = 3;
    static constexpr int n_term = 3;
    static constexpr TermSig term[n_term] = {{GF_T_TIMEOUT, GF_TERM_FLAG_TIME_OUT}, {GF_T_OUT_OF_BOUNDS, 0}, {GF_T_BAD_ORIENTATION, 0}};
    static constexpr int n_rew = 9;
    static constexpr RewSig rew[n_rew] = {{GF_R_CMD_TRACK_LIN_VEL, 0, 0, 0}, {GF_R_CMD_TRACK_ANG_VEL, 0, 0, 2}, {GF_R_LIN_VEL_Z_L2, 0, 0, 0}, {GF_R_ANG_VEL_XY_L2, 0, 0, 0},
                                          {GF_R_HAS_CONTACT, 0, 0, 1}, {GF_R_ACTION_RATE_L2, 0, 0, 0}, {GF_R_DOF_SIMILAR_TO_DEFAULT, 0, 0, 0},
                                          {GF_R_FLAT_ORIENTATION_L2, 0, 0, 0}, {GF_R_TERMINATED, 0, 0, 0}};
    static constexpr int n_cmd = 1;
    static constexpr int cmd_width[GF_POST_MAX_CMD] = {3, 0};
    static constexpr int n_obs = 1;
    static constexpr int obs_width[GF_POST_MAX_OBS] = {48, 0};
    static constexpr int obs_history[GF_POST_MAX_OBS] = {1, 0};
    static constexpr int obs_items[GF_POST_MAX_OBS] = {7, 0};
    static constexpr ItemSig item[GF_POST_MAX_OBS][kPostMaxItems] = {{{GF_O_COMMAND, 3, 0, false, false},
                                                                      {GF_O_ANG_VEL_BODY, 3, 0, false, false},
                                                                      {GF_O_LIN_VEL_BODY, 3, 0, false, false},
                                                                      {GF_O_PROJ_GRAVITY, 3, 0, false, false},
                                                                      {GF_O_DOF_POS, 12, 0, false, false},
                                                                      {GF_O_DOF_VEL, 12, 0, true, false},
                                                                      {GF_O_ACTIONS, 12, 0, false, false}},
                                                                     {}};
    static constexpr int n_air = 1;
    static constexpr int n_gait = 0;
};

// examples/berkeley_humanoid (12 actuated joints, BASELINE config 4): torso contact termination, clamped feet_air_time
struct ProgBerkeleyHumanoid {
    static constexpr bool kStatic = true;
    static constexpr const char* name = "berkeley_humanoid";
    static constexpr int DV = 3;
    static constexpr int n_term = 2;
    static constexpr TermSig term[n_term] = {{GF_T_TIMEOUT, GF_TERM_FLAG_TIME_OUT}, {GF_T_CONTACT_FORCE, 0}};
    static constexpr int n_rew = 7;
    static constexpr RewSig rew[n_rew] = {{GF_R_CMD_TRACK_LIN_VEL, 0, 0, 0}, {GF_R_CMD_TRACK_ANG_VEL, 0, 0, 2}, {GF_R_LIN_VEL_Z_L2, 0, 0, 0}, {GF_R_ANG_VEL_XY_L2, 0, 0, 0},
                                          {GF_R_ACTION_RATE_L2, 0, 0, 0}, {GF_R_DOF_SIMILAR_TO_DEFAULT, 0, 0, 0}, {GF_R_FEET_AIR_TIME, GF_RW_FLAG_MAX, 1, 0}};
    static constexpr int n_cmd = 1;
    static constexpr int cmd_width[GF_POST_MAX_CMD] = {3, 0};
    static constexpr int n_obs = 1;
    static constexpr int obs_width[GF_POST_MAX_OBS] = {48, 0};
    static constexpr int obs_history[GF_POST_MAX_OBS] = {1, 0};
    static constexpr int obs_items[GF_POST_MAX_OBS] = {7, 0};
    static constexpr ItemSig item[GF_POST_MAX_OBS][kPostMaxItems] = {{{GF_O_COMMAND, 3, 0, false, false},
                                                                      {GF_O_ANG_VEL_BODY, 3, 0, false, false},
                                                                      {GF_O_LIN_VEL_BODY, 3, 0, false, false},
                                                                      {GF_O_PROJ_GRAVITY, 3, 0, false, false},
                                                                      {GF_O_DOF_POS, 12, 0, false, false},
                                                                      {GF_O_DOF_VEL, 12, 0, true, false},
                                                                      {GF_O_ACTIONS, 12, 0, false, false}},
                                                                     {}};
    static constexpr int n_air = 1;
    static constexpr int n_gait = 0;
};

// examples/gait_trainer (Go2, BASELINE config 5): velocity + gait command managers (the gait manager stepped / reset in the
// launch, its state row exchanged through LDS), the gait manager's two reward terms on the feet's contact / velocity / position
// buffers, body acceleration, three contact managers, policy (62 x 5) and critic (16 x 5) observations with history
struct ProgGo2GaitTrainer {
    static constexpr bool kStatic = true;
    static constexpr const char* name = "go2_gait_trainer";
    static constexpr int DV = 3;
    static constexpr int n_term = 3;
    static constexpr TermSig term[n_term] = {{GF_T_TIMEOUT, GF_TERM_FLAG_TIME_OUT}, {GF_T_BAD_ORIENTATION, 0}, {GF_T_CONTACT_FORCE, 0}};
    static constexpr int n_rew = 9;
    static constexpr RewSig rew[n_rew] = {{GF_R_GAIT_PHASE, 0, 1, 0}, {GF_R_FOOT_HEIGHT, 0, 1, 0}, {GF_R_BASE_HEIGHT, 0, 0, 0}, {GF_R_CMD_TRACK_LIN_VEL, 0, 1, 0},
                                          {GF_R_CMD_TRACK_ANG_VEL, 0, 1, 2}, {GF_R_BODY_ACCEL_EXP, 0, 0, 0}, {GF_R_LIN_VEL_Z_L2, 0, 0, 0}, {GF_R_ACTION_RATE_L2, 0, 0, 0},
                                          {GF_R_CONTACT_FORCE, 0, 2, 0}};
    static constexpr int n_cmd = 1;
    static constexpr int cmd_width[GF_POST_MAX_CMD] = {3, 0};
    static constexpr int n_obs = 2;
    static constexpr int obs_width[GF_POST_MAX_OBS] = {62, 16};
    static constexpr int obs_history[GF_POST_MAX_OBS] = {5, 5};
    static constexpr int obs_items[GF_POST_MAX_OBS] = {8, 2};
    static constexpr ItemSig item[GF_POST_MAX_OBS][kPostMaxItems] = {{{GF_O_COMMAND, 14, 0, false, false},
                                                                      {GF_O_COMMAND, 3, 1, false, false},
                                                                      {GF_O_ANG_VEL_BODY, 3, 0, false, false},
                                                                      {GF_O_LIN_VEL_BODY, 3, 0, false, false},
                                                                      {GF_O_PROJ_GRAVITY, 3, 0, false, false},
                                                                      {GF_O_DOF_POS, 12, 0, false, false},
                                                                      {GF_O_DOF_VEL, 12, 0, true, false},
                                                                      {GF_O_ACTIONS, 12, 0, false, false}},
                                                                     {{GF_O_CONTACT_FORCE_NORM, 4, 1, false, false}, {GF_O_DOF_FORCE, 12, 0, true, false}}};
    static constexpr int n_air = 0;
    static constexpr int n_gait = 1;
};

// examples/gait_trainer as shipped — with its reset() override (environment.py:347-352) the step is recorded in front of the reset: termination, rewards, command and gait step (GF_POST_NO_RESET) …
struct ProgGo2GaitTrainerFront {
    static constexpr bool kStatic = true;
    static constexpr const char* name = "go2_gait_trainer_front";
    static constexpr int DV = 3;
    static constexpr int n_term = 3;
    static constexpr TermSig term[3] = {{1, 1}, {2, 0}, {6, 0}};
    static constexpr int n_rew = 9;
    static constexpr RewSig rew[9] = {{18, 0, 1, 0}, {19, 0, 1, 0}, {3, 0, 0, 0}, {10, 0, 1, 0}, {11, 0, 1, 2}, {8, 0, 0, 0}, {5, 0, 0, 0}, {9, 0, 0, 0}, {14, 0, 2, 0}};
    static constexpr int n_cmd = 1;
    static constexpr int cmd_width[GF_POST_MAX_CMD] = {3, 0};
    static constexpr int n_obs = 0;
    static constexpr int obs_width[GF_POST_MAX_OBS] = {0, 0};
    static constexpr int obs_history[GF_POST_MAX_OBS] = {0, 0};
    static constexpr int obs_items[GF_POST_MAX_OBS] = {0, 0};
    static constexpr ItemSig item[GF_POST_MAX_OBS][kPostMaxItems] = {
        {},
        {}};
    static constexpr int n_air = 0;
    static constexpr int n_gait = 1;
};

// … and, behind the user's reset, the policy and critic observations (GF_POST_OBSERVE_ONLY); signatures from gf_post_physics_describe
struct ProgGo2GaitTrainerObs {
    static constexpr bool kStatic = true;
    static constexpr const char* name = "go2_gait_trainer_obs";
    static constexpr int DV = 3;
    static constexpr int n_term = 0;
    static constexpr TermSig term[1] = {};
    static constexpr int n_rew = 0;
    static constexpr RewSig rew[1] = {};
    static constexpr int n_cmd = 0;
    static constexpr int cmd_width[GF_POST_MAX_CMD] = {0, 0};
    static constexpr int n_obs = 2;
    static constexpr int obs_width[GF_POST_MAX_OBS] = {62, 16};
    static constexpr int obs_history[GF_POST_MAX_OBS] = {5, 5};
    static constexpr int obs_items[GF_POST_MAX_OBS] = {8, 2};
    static constexpr ItemSig item[GF_POST_MAX_OBS][kPostMaxItems] = {
        {{1, 14, 0, false, false}, {1, 3, 1, false, false}, {2, 3, 0, false, false}, {3, 3, 0, false, false}, {4, 3, 0, false, false}, {5, 12, 0, false, false}, {6, 12, 0, true, false}, {8, 12, 0, false, false}},
        {{10, 4, 0, false, false}, {7, 12, 0, true, false}}};
    static constexpr int n_air = 0;
    static constexpr int n_gait = 0;
    static constexpr bool term_done = true;   // the termination masks are inputs (Python-level terms ran behind a termination launch of its own)
};

// ---- matching -------------------------------------------------------------------------------------------------------------------
// a recorded signature (registered with tools/register_program.py)
struct ProgHumanoid28Stress {
    static constexpr bool kStatic = true;
    static constexpr const char* name = "humanoid28_stress";
    static constexpr int DV = 7;
    static constexpr int n_term = 4;
    static constexpr TermSig term[4] = {{1, 1}, {2, 0}, {6, 0}, {3, 0}};
    static constexpr int n_rew = 12;
    static constexpr RewSig rew[12] = {{10, 0, 0, 0}, {11, 0, 0, 2}, {8, 0, 0, 0}, {9, 0, 0, 0}, {4, 0, 0, 0}, {12, 0, 0, 0}, {15, 4, 1, 0}, {14, 0, 2, 0}, {16, 0, 1, 0}, {1, 0, 0, 0}, {7, 0, 0, 0}, {3, 0, 0, 0}};
    static constexpr int n_cmd = 2;
    static constexpr int cmd_width[GF_POST_MAX_CMD] = {3, 1};
    static constexpr int n_obs = 2;
    static constexpr int obs_width[GF_POST_MAX_OBS] = {94, 61};
    static constexpr int obs_history[GF_POST_MAX_OBS] = {3, 1};
    static constexpr int obs_items[GF_POST_MAX_OBS] = {7, 4};
    static constexpr ItemSig item[GF_POST_MAX_OBS][kPostMaxItems] = {
        {{1, 1, 1, false, true}, {1, 3, 0, false, true}, {2, 3, 0, true, true}, {4, 3, 0, false, true}, {5, 28, 0, false, true}, {6, 28, 0, true, true}, {8, 28, 0, false, true}},
        {{10, 2, 1, true, false}, {3, 3, 0, false, false}, {7, 28, 0, false, false}, {9, 28, 0, false, false}}};
    static constexpr int n_air = 2;
    static constexpr int n_gait = 0;
};

// dynamic LDS of post_ws_kernel<P>, in floats (a static program keeps no copy of the descriptor in LDS; the interpreter adds one)
template <class P>
inline size_t lds_ws_floats(int omax, int n_gait) {
    return (size_t)(x_fields(n_gait) + ws_sum_rows<P>() + ws_aux_rows<P>() + ws_pre_rows<P>() + ws_obs_norm_rows<P>()) * kEnvBlock + (size_t)(omax + 1) * kEnvBlock;
}

template <class P>
bool program_matches(const GfPostArgs& a) {
    if ((a.term_done != 0) != prog_term_done<P>::value || a.num_dofs != 4 * P::DV || a.num_term != P::n_term || (a.num_rew < 0 ? 0 : a.num_rew) != P::n_rew || a.n_cmd != P::n_cmd || a.n_obs != P::n_obs || a.n_air != P::n_air ||
        a.n_gait != P::n_gait)
        return false;
    for (int k = 0; k < P::n_term; ++k)
        if (a.tterms[k].op != P::term[k].op || a.tterms[k].flags != P::term[k].flags) return false;
    for (int k = 0; k < P::n_rew; ++k) {
        const GfTerm& t = a.rterms[k];
        if (t.op != P::rew[k].op || t.flags != P::rew[k].flags || t.i[0] != P::rew[k].i0 || t.i[1] != P::rew[k].i1) return false;
    }
    for (int c = 0; c < P::n_cmd; ++c)
        if (a.cmds[c].width != P::cmd_width[c]) return false;
    for (int m = 0; m < P::n_obs; ++m) {
        const PostObs& ob = a.obs[m];
        if (ob.width != P::obs_width[m] || ob.history != P::obs_history[m] || ob.num_items != P::obs_items[m]) return false;
        for (int i = 0; i < P::obs_items[m]; ++i) {
            const GfObsItem& it = ob.items[i];
            const ItemSig& sg = P::item[m][i];
            if (it.op != sg.op || it.width != sg.width || it.i0 != sg.i0 || (it.scale != 1.0f) != sg.scaled || (it.noise != 0.0f) != sg.noisy) return false;
        }
    }
    return true;
}

// prints the signature of a packed descriptor as a program struct (the notation above)
inline int describe_program(const GfPostArgs& a, char* buf, int cap) {
    int n = 0;
    auto put = [&](const char* fmt, auto... v) {
        if (n < cap) {
            const int w = snprintf(buf + n, (size_t)(cap - n), fmt, v...);
            n += w > 0 ? w : 0;
        }
    };
    put("DV = %d; n_term = %d; term = {", (a.num_dofs + 3) / 4, a.num_term);
    for (int k = 0; k < a.num_term; ++k) put("{%d, %d}%s", a.tterms[k].op, a.tterms[k].flags, k + 1 < a.num_term ? ", " : "");
    put("}; n_rew = %d; rew = {", a.num_rew);
    for (int k = 0; k < a.num_rew; ++k) put("{%d, %d, %d, %d}%s", a.rterms[k].op, a.rterms[k].flags, a.rterms[k].i[0], a.rterms[k].i[1], k + 1 < a.num_rew ? ", " : "");
    put("}; n_cmd = %d; cmd_width = {", a.n_cmd);
    for (int c = 0; c < a.n_cmd; ++c) put("%d%s", a.cmds[c].width, c + 1 < a.n_cmd ? ", " : "");
    put("}; n_obs = %d;", a.n_obs);
    for (int m = 0; m < a.n_obs; ++m) {
        put(" obs[%d]: width %d history %d items {", m, a.obs[m].width, a.obs[m].history);
        for (int i = 0; i < a.obs[m].num_items; ++i) {
            const GfObsItem& it = a.obs[m].items[i];
            put("{%d, %d, %d, %s, %s}%s", it.op, it.width, it.i0, it.scale != 1.0f ? "true" : "false", it.noise != 0.0f ? "true" : "false", i + 1 < a.obs[m].num_items ? ", " : "");
        }
        put("%s", "};");
    }
    put(" n_air = %d; n_gait = %d", a.n_air, a.n_gait);
    if (a.term_done) put("%s", "; term_done = 1");
    return n;
}

}  // namespace gf
