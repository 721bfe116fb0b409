"""Clock / power / temperature sampling around a timed region (bench.py).

MI355X clocks to its power budget (MI355X_MICROARCH.md, "DVFS give-back"): a dense-MFMA kernel runs at whatever shader
clock the socket power limit allows, so the same binary is faster in a short burst than in a sustained run and differs from
box to box.  A throughput number therefore travels with the clocks and the power it was measured at.  This module samples
them from a background thread (amdsmi's gpu-metrics table; `rocm-smi --json` as a fallback) and never raises: telemetry
that cannot be read is reported as {"available": false, "why": ...}.
"""
import json
import subprocess
import threading
import time

_KEYS = ("current_gfxclk", "average_gfxclk_frequency", "current_uclk", "current_socclk", "average_socket_power",
         "current_socket_power", "temperature_hotspot", "temperature_mem", "average_gfx_activity", "average_umc_activity",
         "throttle_status", "indep_throttle_status", "accumulated_counter", "prochot_residency_acc", "ppt_residency_acc",
         "socket_thm_residency_acc", "vr_thm_residency_acc", "hbm_thm_residency_acc", "gfxclk_lock_status", "energy_accumulator")


def _num(v):
    if isinstance(v, bool):
        return int(v)
    if isinstance(v, (int, float)):
        return v
    return None


class _Amdsmi:
    def __init__(self, index):
        import amdsmi
        self.m = amdsmi
        amdsmi.amdsmi_init()
        hs = amdsmi.amdsmi_get_processor_handles()
        self.h = hs[index if index < len(hs) else 0]
        self.sample()                                       # fail here, not in the thread

    def sample(self):
        d = self.m.amdsmi_get_gpu_metrics_info(self.h)
        out = {}
        for k in _KEYS:
            v = _num(d.get(k))
            if v is not None and v not in (0xFFFF, 0xFFFFFFFF, 0xFFFFFFFFFFFFFFFF):
                out[k] = v
        xs = [x for x in (d.get("current_gfxclks") or []) if isinstance(x, (int, float)) and 0 < x < 0xFFFF]
        if xs:
            out["gfxclk_xcd_min"], out["gfxclk_xcd_max"] = min(xs), max(xs)
            out["gfxclk_xcd_mean"] = sum(xs) / len(xs)
        return out

    def static(self):
        out = {}
        try:
            c = self.m.amdsmi_get_power_cap_info(self.h)
            out["power_cap_W"] = {k: _num(v) for k, v in c.items() if _num(v) is not None}
        except Exception as e:                              # noqa: BLE001
            out["power_cap_W"] = repr(e)[:80]
        for name, typ in (("gfx", "GFX"), ("mem", "MEM")):
            try:
                c = self.m.amdsmi_get_clock_info(self.h, getattr(self.m.AmdSmiClkType, typ))
                out[name + "_clock_MHz"] = {k: _num(v) for k, v in c.items() if _num(v) is not None}
            except Exception as e:                          # noqa: BLE001
                out[name + "_clock_MHz"] = repr(e)[:80]
        return out


class _RocmSmi:
    def __init__(self, index):
        self.i = index
        self.sample()

    def sample(self):
        r = subprocess.run(["rocm-smi", "-d", str(self.i), "--showclocks", "--showpower", "--showtemp", "--json"],
                           capture_output=True, text=True, timeout=10)
        d = json.loads(r.stdout)
        card = next(iter(d.values()))
        out = {}
        for k, v in card.items():
            kl = k.lower()
            try:
                if "sclk" in kl and "level" in kl:
                    out["current_gfxclk"] = float(str(v).strip("()").replace("Mhz", "").replace("MHz", ""))
                elif "power" in kl and "(w)" in kl:
                    out["current_socket_power"] = float(v)
                elif "junction" in kl or "hotspot" in kl:
                    out["temperature_hotspot"] = float(v)
            except ValueError:
                pass
        return out

    def static(self):
        return {}


class Telemetry:
    """t = Telemetry(device_index); t.start(); ...timed region...; t.stop(); t.report() -> JSON-able dict."""

    def __init__(self, index=0, period_s=0.02):
        self.period = period_s
        self.src, self.why = None, None
        for cls in (_Amdsmi, _RocmSmi):
            try:
                self.src = cls(index)
                self.kind = cls.__name__.strip("_").lower()
                if cls is _RocmSmi:
                    self.period = max(period_s, 0.25)       # a subprocess per sample
                break
            except Exception as e:                          # noqa: BLE001
                self.why = f"{cls.__name__}: {e!r}"[:200]
        self.series, self.marks = [], {}
        self._stop, self._thr = threading.Event(), None

    def _snap(self):
        try:
            s = self.src.sample()
        except Exception:                                   # noqa: BLE001
            return None
        s["t"] = time.perf_counter()
        return s

    def start(self):
        if self.src is None or self._thr is not None:
            return
        self.idle = self._snap()

        def loop():
            while not self._stop.is_set():
                s = self._snap()
                if s:
                    self.series.append(s)
                self._stop.wait(self.period)

        self._thr = threading.Thread(target=loop, daemon=True)
        self._thr.start()

    def mark(self, name):
        self.marks[name] = time.perf_counter()

    def stop(self):
        if self._thr is None:
            return
        self._stop.set()
        self._thr.join(timeout=5)

    def report(self, begin="timed_begin", end="timed_end"):
        if self.src is None:
            return {"available": False, "why": self.why}
        t0, t1 = self.marks.get(begin), self.marks.get(end)
        ins = [s for s in self.series if t0 is not None and t1 is not None and t0 <= s["t"] <= t1]
        pre = [s for s in self.series if t0 is not None and s["t"] < t0]

        def stats(rows, key):
            v = [r[key] for r in rows if key in r]
            if not v:
                return None
            v = sorted(v)
            return {"min": v[0], "median": v[len(v) // 2], "max": v[-1], "n": len(v)}

        out = {"available": True, "source": self.kind, "period_ms": self.period * 1e3, "samples_in_timed_region": len(ins),
               "before_any_work": {k: v for k, v in (self.idle or {}).items() if k != "t"}}
        try:
            out["static"] = self.src.static()
        except Exception:                                   # noqa: BLE001
            pass
        for key, label in (("current_gfxclk", "gfxclk_MHz"), ("gfxclk_xcd_mean", "gfxclk_xcd_mean_MHz"),
                           ("gfxclk_xcd_min", "gfxclk_xcd_min_MHz"), ("current_uclk", "uclk_MHz"),
                           ("current_socket_power", "socket_power_W"), ("average_socket_power", "avg_socket_power_W"),
                           ("temperature_hotspot", "hotspot_C"), ("temperature_mem", "hbm_C")):
            for name, rows in (("timed", ins), ("warmup", pre)):
                s = stats(rows, key)
                if s:
                    out.setdefault(name, {})[label] = s
        # which limiter was active: residency accumulators advance while that limiter throttles the clocks
        if ins:
            first, last = (pre or ins)[0], ins[-1]
            acc = {}
            for k in ("accumulated_counter", "ppt_residency_acc", "prochot_residency_acc", "socket_thm_residency_acc",
                      "vr_thm_residency_acc", "hbm_thm_residency_acc", "energy_accumulator"):
                if k in first and k in last:
                    acc[k] = last[k] - first[k]
            out["limiter_residency_delta"] = acc
            ts = [s.get("throttle_status") for s in ins if "throttle_status" in s]
            if ts:
                out["throttle_status_any"] = int(any(ts))
            # coarse time series (<= 40 points) of clock and power across warm-up + timed region
            rows = pre + ins
            stride = max(1, len(rows) // 40)
            base = rows[0]["t"]
            out["series"] = [[round((r["t"] - base) * 1e3), r.get("gfxclk_xcd_mean", r.get("current_gfxclk")),
                              r.get("current_socket_power", r.get("average_socket_power")), r.get("temperature_hotspot")]
                             for r in rows[::stride]]
            out["series_columns"] = ["ms", "gfxclk_MHz", "socket_power_W", "hotspot_C"]
            out["timed_region_starts_at_ms"] = round((t0 - base) * 1e3) if t0 else None
        return out
