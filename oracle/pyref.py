"""A SECOND, independent CPU restatement of the reference's algorithm, in pure Python -- TEST INFRASTRUCTURE.

Why it exists.  oracle/lbfgs_oracle.c is pinned to the reference's own known answers (tests/test_oracle_golden.py), but those
cover the default path only: Powell damping, `gradient_only`, the three backtracking variants without OWL-QN and m != 6 are
held by no test or vector of the reference -- the C oracle follows the source text there and nothing
checks the transcription.  This module restates the same source a second time, written from the Rust text alone (not from the
C oracle), in another language and another shape (lists of Python floats, no shared helper), and
tests/test_oracle_pyref.py requires the two to agree BIT FOR BIT on every iteration of runs that take exactly those paths.
Two independent readings that agree to the last bit do not make a reference vector, but a slip in either transcription --
an argument order in MCSTEP, a stale `ys`, a `<` for a `<=` -- shows up as a difference.

Arithmetic: Python floats are IEEE f64; `y + c * x` rounds twice (no FMA); sums run left to right from 0.0 -- the order of
src/math.rs:31-82.  Only small problems (pure-Python loops).

Every function cites the lines it restates.  Nothing here is imported by the product, by bench.py or by the C oracle.
"""
import math

INF = float("inf")


# ------------------------------------------------------------------------------------------------ src/math.rs:31-82
def vecadd(y, x, c):          # :32-36   y += c*x
    for i in range(min(len(y), len(x))):
        y[i] = y[i] + c * x[i]


def vecdot(x, y):             # :38-40   left-to-right sum of the products
    s = 0.0
    for a, b in zip(x, y):
        s = s + a * b
    return s


def vecscale(y, c):           # :42-46
    for i in range(len(y)):
        y[i] = y[i] * c


def veccpy(y, x):             # :48-52
    for i in range(min(len(y), len(x))):
        y[i] = x[i]


def vecncpy(y, x):            # :54-58
    for i in range(min(len(y), len(x))):
        y[i] = -x[i]


def vecdiff(z, x, y):         # :60-64   z = x - y
    for i in range(min(len(z), len(x), len(y))):
        z[i] = x[i] - y[i]


def vec2norm(x):              # :66-69
    return math.sqrt(vecdot(x, x))


def vec2norminv(x):           # :71-73
    return 1.0 / vec2norm(x)


class LbfgsError(Exception):
    pass


# ------------------------------------------------------------------------------------------------ src/orthantwise.rs
def signum(v):                # :174-180: NaN and +-0 -> 0
    if v != v or v == 0.0:
        return 0.0
    return math.copysign(1.0, v)


class Orthantwise:            # :33-55
    def __init__(self, c=1.0, start=0, end=None):
        self.c, self.start, self.end = c, start, end

    def start_end(self, x):   # :59-67
        n = len(x)
        end = min(n if self.end is None else self.end, n)
        if not self.start < end:
            raise AssertionError("invalid start for orthantwise")
        return self.start, end

    def x1norm(self, x):      # :70-79: c * |x_i| inside the sum
        lo, hi = self.start_end(x)
        s = 0.0
        for i in range(lo, hi):
            s = s + self.c * abs(x[i])
        return s

    def compute_pseudo_gradient(self, pg, x, g):   # :82-112
        lo, hi = self.start_end(x)
        for i in range(0, lo):
            pg[i] = g[i]
        c = self.c
        for i in range(lo, hi):
            if x[i] != 0.0:
                pg[i] = g[i] + math.copysign(1.0, x[i]) * c if x[i] == x[i] else g[i] + x[i] * c   # f64::signum(NaN) = NaN
            else:
                right, left = g[i] + c, g[i] - c
                if right < 0.0:
                    pg[i] = right
                elif left > 0.0:
                    pg[i] = left
                else:
                    pg[i] = 0.0
        for i in range(hi, len(g)):
            pg[i] = g[i]

    def constraint_line_search(self, x, wp):        # :118-133 with project :165-171
        lo, hi = self.start_end(x)
        for i in range(lo, hi):
            if signum(x[i]) != signum(wp[i]):
                x[i] = 0.0

    def constrain_search_direction(self, d, pg):    # :140-161
        lo, hi = self.start_end(pg)
        for i in range(lo, hi):
            if signum(d[i]) != signum(-pg[i]):
                d[i] = 0.0
        if vec2norm(d) == 0.0:
            raise AssertionError("invalid direction vector after constraints")


# ------------------------------------------------------------------------------------------------ src/core.rs:10-218
class Problem:
    def __init__(self, x, evaluate, owlqn):         # :59-75
        n = len(x)
        self.x, self.fx = x, 0.0
        self.gx, self.xp, self.gp = [0.0] * n, [0.0] * n, [0.0] * n
        self.pg, self.wp, self.d = [0.0] * n, [0.0] * n, [0.0] * n
        self.eval_fn, self.owlqn, self.neval = evaluate, owlqn, 0

    def dginit(self):                               # :78-92 (a positive value only warns)
        return vecdot(self.gx, self.d) if self.owlqn is None else vecdot(self.pg, self.d)

    def update_search_direction(self):              # :95-101
        vecncpy(self.d, self.pg if self.owlqn is not None else self.gx)

    def dg_unchecked(self):                         # :114-116: the RAW gradient, OWL-QN or not
        return vecdot(self.gx, self.d)

    def evaluate(self):                             # :119-132
        self.fx = self.eval_fn(self.x, self.gx)
        if self.owlqn is not None:
            self.fx = self.fx + self.owlqn.x1norm(self.x)
            self.owlqn.compute_pseudo_gradient(self.pg, self.x, self.gx)
        self.neval += 1

    def take_line_step(self, step):                 # :155-164
        veccpy(self.x, self.xp)
        vecadd(self.x, self.d, step)
        if self.owlqn is not None:
            self.owlqn.constraint_line_search(self.x, self.wp)

    def update_orthant_new_point(self):             # :167-180, over ALL i
        for i in range(len(self.x)):
            self.wp[i] = signum(-self.pg[i]) if self.xp[i] == 0.0 else signum(self.xp[i])

    def gnorm(self):                                # :183-189
        return vec2norm(self.pg) if self.owlqn is not None else vec2norm(self.gx)

    def xnorm(self):                                # :192-194
        return vec2norm(self.x)

    def revert(self):                               # :201-204: x and gx only -- NOT fx, NOT pg
        veccpy(self.x, self.xp)
        veccpy(self.gx, self.gp)

    def save_state(self):                           # :207-210
        veccpy(self.xp, self.x)
        veccpy(self.gp, self.gx)

    def constrain_search_direction(self):           # :213-217
        if self.owlqn is not None:
            self.owlqn.constrain_search_direction(self.d, self.pg)


# ------------------------------------------------------------------------------------------------ src/line.rs
class LineSearch:             # :151-162
    def __init__(self):
        self.ftol, self.gtol, self.xtol = 1e-4, 0.9, 2.220446049250313e-16
        self.min_step, self.max_step, self.max_linesearch = 1e-20, 1e20, 20
        self.gradient_only, self.algorithm = False, "MoreThuente"

    def validate_step(self, step):                  # :166-177
        if step < self.min_step:
            raise LbfgsError("The line-search step became smaller than LineSearch::min_step.")
        if step > self.max_step:
            raise LbfgsError("The line-search step became larger than LineSearch::max_step.")

    def find(self, prb, step):                      # :193-223 -> (number of calls, step)
        if math.copysign(1.0, step) < 0.0:
            raise LbfgsError("A logic error (negative line-search step) occurred.")
        if self.algorithm == "MoreThuente" and prb.owlqn is None and self.gradient_only:
            raise LbfgsError("Gradient only optimization is incompatible with MoreThuente line search.")
        box = [step]
        try:
            if self.algorithm == "MoreThuente" and prb.owlqn is None:
                n = morethuente(prb, box, self)
            else:
                n = backtracking(prb, box, self)
        except LbfgsError:                           # swallowed: :213-220
            prb.revert()
            n = 0
        return n, box[0]


def cubic_minimizer(u, fu, du, v, fv, dv):          # :620-637 (no guard under the root)
    d = v - u
    theta = (fu - fv) * 3.0 / d + du + dv
    p, q, r = abs(theta), abs(du), abs(dv)
    s = max(max(p, q), r)
    a = theta / s
    gamma = s * math.sqrt(a * a - du / s * (dv / s))
    if v < u:
        gamma = -gamma
    p = gamma - du + theta
    q = gamma - du + gamma + dv
    r = p / q
    return u + r * d


def cubic_minimizer2(u, fu, du, v, fv, dv, xmin, xmax):   # :653-677
    d = v - u
    theta = (fu - fv) * 3.0 / d + du + dv
    p, q, r = abs(theta), abs(du), abs(dv)
    s = max(max(p, q), r)
    a = theta / s
    gamma = s * math.sqrt(max(0.0, a * a - du / s * (dv / s)))
    if u < v:
        gamma = -gamma
    p = gamma - dv + theta
    q = gamma - dv + gamma + du
    r = p / q
    if r < 0.0 and gamma != 0.0:
        return v - r * d
    return xmax if v > u else xmin


def quard_minimizer(u, fu, du, v, fv):              # :689-692
    a = v - u
    return u + du / ((fu - fv) / a + du) / 2.0 * a


def quard_minimizer2(u, du, v, dv):                 # :705-708
    a = u - v
    return v + dv / (dv - du) * a


def update_trial_interval(S, ft, dt, tmin, tmax):   # :446-606.  S: dict x, fx, dx, y, fy, dy, t, brackt -- updated in place
    x, fx, dx, y, fy, dy, t = S["x"], S["fx"], S["dx"], S["y"], S["fy"], S["dy"], S["t"]
    dsign = dt * (dx / abs(dx)) < 0.0                # :461
    if S["brackt"]:                                  # :467-477
        if t <= min(x, y) or max(x, y) <= t:
            raise LbfgsError("The line-search step went out of the interval of uncertainty.")
        if 0.0 <= dx * (t - x):
            raise LbfgsError("The current search direction increases the objective function value.")
        if tmax < tmin:
            raise LbfgsError("A logic error occurred; alternatively, the interval of uncertainty became too small.")
    if fx < ft:                                      # case 1 :481-496
        S["brackt"] = True
        mc = cubic_minimizer(x, fx, dx, t, ft, dt)
        mq = quard_minimizer(x, fx, dx, t, ft)
        newt = mc if abs(mc - x) < abs(mq - x) else mc + 0.5 * (mq - mc)
        bound = 1
    elif dsign:                                      # case 2 :497-511
        S["brackt"] = True
        mc = cubic_minimizer(x, fx, dx, t, ft, dt)
        mq = quard_minimizer2(x, dx, t, dt)
        newt = mc if abs(mc - t) > abs(mq - t) else mq
        bound = 0
    elif abs(dt) < abs(dx):                          # case 3 :512-537
        mc = cubic_minimizer2(x, fx, dx, t, ft, dt, tmin, tmax)
        mq = quard_minimizer2(x, dx, t, dt)
        if S["brackt"]:
            newt = mc if abs(t - mc) < abs(t - mq) else mq
        else:
            newt = mc if abs(t - mc) > abs(t - mq) else mq
        bound = 1
    else:                                            # case 4 :538-553
        if S["brackt"]:
            newt = cubic_minimizer(t, ft, dt, y, fy, dy)
        elif x < t:
            newt = tmax
        else:
            newt = tmin
        bound = 0
    if fx < ft:                                      # :564-580
        y, fy, dy = t, ft, dt
    else:
        if dsign:
            y, fy, dy = x, fx, dx
        x, fx, dx = t, ft, dt
    if tmax < newt:                                  # :583-588
        newt = tmax
    if newt < tmin:
        newt = tmin
    if S["brackt"] and bound != 0:                   # :592-601
        mq = x + 0.66 * (y - x)
        if x < y:
            if mq < newt:
                newt = mq
        elif newt < mq:
            newt = mq
    S.update(x=x, fx=fx, dx=dx, y=y, fy=fy, dy=dy, t=newt)
    return 0                                         # :605: always Ok(0)


def morethuente(prb, stp, p):                        # :226-399.  stp: one-element list (in / out)
    dginit = prb.dginit()
    brackt, stage1, uinfo = False, 1, 0
    finit = prb.fx
    dgtest = p.ftol * dginit
    width = p.max_step - p.min_step
    prev_width = 2.0 * width
    stx = sty = 0.0
    fx = fy = finit
    dgx = dgy = dginit
    for count in range(1, p.max_linesearch):         # :258: 1 .. max_linesearch - 1
        if brackt:
            stmin, stmax = (stx if stx <= sty else sty), (stx if stx >= sty else sty)
        else:
            stmin, stmax = stx, stp[0] + 4.0 * (stp[0] - stx)
        if stp[0] < p.min_step:
            stp[0] = p.min_step
        if p.max_step < stp[0]:
            stp[0] = p.max_step
        if (brackt and (stp[0] <= stmin or stmax <= stp[0] or p.max_linesearch <= count + 1 or uinfo != 0)) or \
                (brackt and stmax - stmin <= p.xtol * stmax):
            stp[0] = stx
        prb.take_line_step(stp[0])
        prb.evaluate()
        f = prb.fx
        dg = prb.dg_unchecked()
        ftest1 = finit + stp[0] * dgtest
        if brackt and (stp[0] <= stmin or stmax <= stp[0] or uinfo != 0):
            raise LbfgsError("A rounding error occurred")
        if brackt and stmax - stmin <= p.xtol * stmax:
            raise LbfgsError("Relative width of the interval of uncertainty is at most xtol.")
        if stp[0] == p.max_step and f <= ftest1 and dg <= dgtest:
            raise LbfgsError("The line-search step became larger than LineSearch::max_step.")
        if stp[0] == p.min_step and (ftest1 < f or dgtest <= dg):
            raise LbfgsError("The line-search step became smaller than LineSearch::min_step.")
        if abs(dg) <= p.gtol * -dginit:              # :315-317: the curvature condition alone
            return count
        if stage1 != 0 and f <= ftest1 and min(p.ftol, p.gtol) * dginit <= dg:
            stage1 = 0
        if stage1 != 0 and ftest1 < f and f <= fx:   # :335-367: the modified function
            S = dict(x=stx, fx=fx - stx * dgtest, dx=dgx - dgtest, y=sty, fy=fy - sty * dgtest, dy=dgy - dgtest, t=stp[0], brackt=brackt)
            uinfo = update_trial_interval(S, f - stp[0] * dgtest, dg - dgtest, stmin, stmax)
            stx, sty, stp[0], brackt = S["x"], S["y"], S["t"], S["brackt"]
            fx = S["fx"] + stx * dgtest
            fy = S["fy"] + sty * dgtest
            dgx = S["dx"] + dgtest
            dgy = S["dy"] + dgtest
        else:
            S = dict(x=stx, fx=fx, dx=dgx, y=sty, fy=fy, dy=dgy, t=stp[0], brackt=brackt)
            uinfo = update_trial_interval(S, f, dg, stmin, stmax)
            stx, fx, dgx, sty, fy, dgy, stp[0], brackt = S["x"], S["fx"], S["dx"], S["y"], S["fy"], S["dy"], S["t"], S["brackt"]
        if not brackt:
            continue
        if 0.66 * prev_width <= abs(sty - stx):
            stp[0] = stx + 0.5 * (sty - stx)
        prev_width = width
        width = abs(sty - stx)
    return p.max_linesearch                          # :398


def backtracking(prb, stp, p):                       # :716-784
    dginit = prb.dginit()
    dec, inc = 0.5, 2.1
    finit = prb.fx
    dgtest = p.ftol * dginit
    owl = prb.owlqn is not None
    if owl:
        prb.update_orthant_new_point()               # :735
    for count in range(1, p.max_linesearch):
        prb.take_line_step(stp[0])
        prb.evaluate()
        if prb.fx > finit + stp[0] * dgtest:
            width = dec
        elif p.algorithm == "BacktrackingArmijo" or owl:
            return count
        else:
            dg = prb.dg_unchecked()
            if dg < p.gtol * dginit:
                width = inc
            elif p.algorithm == "BacktrackingWolfe":
                return count
            elif dg > -p.gtol * dginit:
                width = dec
            else:
                return count
        if p.gradient_only:                          # :768-774
            dg = prb.dg_unchecked()
            if abs(dg) <= -p.gtol * abs(dginit):
                return count
        p.validate_step(stp[0])                      # :776: AFTER the tests, before the step changes
        stp[0] = stp[0] * width
    return p.max_linesearch


# ------------------------------------------------------------------------------------------------ src/lbfgs.rs
class IterationData:          # :607-627
    def __init__(self, n):
        self.alpha, self.ys, self.s, self.y = 0.0, 0.0, [0.0] * n, [0.0] * n

    def update(self, x, xp, gx, gp, step, damping):   # :640-692
        vecdiff(self.s, x, xp)
        if vec2norm(self.s) == 0.0:
            raise LbfgsError("x not changed")
        vecdiff(self.y, gx, gp)
        ys = vecdot(self.y, self.s)
        yy = vecdot(self.y, self.y)
        if yy == 0.0:
            raise LbfgsError("gx not changed")
        self.ys = ys                                 # stored BEFORE damping, never refreshed (:656)
        sigma2, sigma3 = 0.6, 3.0
        if damping:
            bs = list(gp)
            vecscale(bs, -step)
            sbs = vecdot(self.s, bs)
            if ys < (1.0 - sigma2) * sbs:            # case 1: y is replaced
                theta = sigma2 * sbs / (sbs - ys)
                vecscale(bs, 1.0 - theta)
                vecadd(bs, self.y, theta)
                veccpy(self.y, bs)
            elif ys > (1.0 + sigma3) * sbs:          # case 2: computed and dropped (:681-685)
                theta = sigma3 * sbs / (ys - sbs)
                vecscale(bs, 1.0 - theta)
                vecadd(bs, self.y, theta)
        return ys / yy                               # :691


def two_loop_recursion(lm, d, gamma, m, k, end):     # :569-604
    end = (end + 1) % m
    j = end
    bound = min(m, k)
    for _ in range(bound):
        j = (j + m - 1) % m
        it = lm[j]
        it.alpha = vecdot(it.s, d) / it.ys
        vecadd(d, it.y, -it.alpha)
    vecscale(d, gamma)
    for _ in range(bound):
        it = lm[j]
        beta = vecdot(it.y, d) / it.ys
        vecadd(d, it.s, it.alpha - beta)
        j = (j + 1) % m
    return end


class Lbfgs:                  # :161-176 defaults, :185-384 setters (only what the tests use)
    def __init__(self):
        self.m, self.epsilon, self.max_iterations, self.max_evaluations = 6, 1e-5, 0, 0
        self.orthantwise, self.linesearch = None, LineSearch()
        self.initial_inverse_hessian, self.max_step_size, self.damping, self.constrain_step_size = 1.0, 1.0, False, True

    def with_gradient_only(self):                    # :283-289
        self.linesearch.gradient_only = True
        self.damping = True
        self.linesearch.algorithm = "BacktrackingStrongWolfe"
        return self

    def minimize(self, x, evaluate, progress=None):  # :399-421 -> (rows, error or None)
        rows = []
        try:
            st = State(self, x, evaluate)            # :443-481
            while True:
                if st.is_converged():
                    break
                p = st.propagate()
                rows.append(p)
                if progress is not None and progress(p):
                    break
        except LbfgsError as e:
            return rows, str(e)
        return rows, None


class State:
    def __init__(self, par, x, evaluate):            # build: :443-481
        self.v = par
        self.lm = [IterationData(len(x)) for _ in range(par.m)]
        self.prb = Problem(x, evaluate, par.orthantwise)
        self.prb.evaluate()
        self.prb.update_search_direction()
        self.step = vec2norminv(self.prb.d) * par.initial_inverse_hessian
        self.end, self.k, self.ncall = 0, 0, 0

    def progress(self):                              # core.rs:253-268
        return dict(niter=self.k, neval=self.prb.neval, ncall=self.ncall, fx=self.prb.fx, xnorm=self.prb.xnorm(),
                    gnorm=self.prb.gnorm(), step=self.step)

    def is_converged(self):                          # :489-494, :695-748
        p, v = self.progress(), self.v
        if v.max_iterations != 0 and p["niter"] >= v.max_iterations:
            return True
        if v.max_evaluations != 0 and p["neval"] >= v.max_evaluations:
            return True
        return p["gnorm"] / max(p["xnorm"], 1.0) <= v.epsilon

    def propagate(self):                             # :503-560
        self.k += 1
        if self.k == 1:
            return self.progress()
        prb, v = self.prb, self.v
        prb.save_state()
        self.ncall, self.step = v.linesearch.find(prb, self.step)
        step_ls = self.step
        gamma = self.lm[self.end].update(prb.x, prb.xp, prb.gx, prb.gp, self.step, v.damping)
        prb.update_search_direction()
        self.end = two_loop_recursion(self.lm, prb.d, gamma, v.m, self.k - 1, self.end)
        dnorm = vec2norm(prb.d)
        if math.copysign(1.0, dnorm) < 0.0:          # :544 is_sign_positive (a NaN with a clear sign bit passes, as in Rust)
            raise LbfgsError("invalid norm value")
        if v.constrain_step_size:
            self.step = min(v.max_step_size, dnorm) / dnorm
        else:
            self.step = 1.0
        prb.constrain_search_direction()
        p = self.progress()
        p["step"] = step_ls
        return p
