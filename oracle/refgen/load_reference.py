"""Import the unmodified Python reference from /root/reference -- TEST INFRASTRUCTURE ONLY.

Only usable in the build container (the reference never travels to the GPU
box).  One preset per process: the reference's constants are module globals
(RR_Constants.py:4 `GAME_MODE`), so preset T is obtained by exec-ing the text of
RR_Constants.py with that one assignment flipped into a module object that is
registered before anything else imports it.  Preset D (the 1+1 / 1+1 duel) keeps
GAME_MODE = True and sets the four entity counts of RR_Constants.py:30-34 to 1 the
same way; preset X sets them to 2 + 1 robots and 2 + 3 balls (a shape outside the
library's built list: odd counts, unequal teams).  Nothing on disk changes.
"""
import os
import sys
import types

REF_ROOT = os.environ.get("RR_REFERENCE_ROOT", "/root/reference")


def reference_available():
    return os.path.isfile(os.path.join(REF_ROOT, "MyUtils.py"))


def load_reference(preset):
    """Returns a namespace with the reference modules for preset 'G' or 'T'."""
    assert preset in ("G", "T", "D", "X")
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REF_ROOT)
    sys.dont_write_bytecode = True
    here = os.path.dirname(os.path.abspath(__file__))
    if here not in sys.path:
        sys.path.insert(0, here)
    import standins
    standins.install()
    if REF_ROOT not in sys.path:
        sys.path.insert(0, REF_ROOT)
    if "robo_rugby.gym_env.RR_Constants" in sys.modules:
        const = sys.modules["robo_rugby.gym_env.RR_Constants"]
        have = {4: "G", 2: "D", 3: "X"}[const.NUM_ROBOTS_TOTAL] if const.GAME_MODE else "T"
        if have != preset:
            raise RuntimeError("reference already loaded with preset %s in this process" % have)
    elif preset in ("T", "D", "X"):
        pkg = types.ModuleType("robo_rugby")
        pkg.__path__ = [os.path.join(REF_ROOT, "robo_rugby")]
        sub = types.ModuleType("robo_rugby.gym_env")
        sub.__path__ = [os.path.join(REF_ROOT, "robo_rugby", "gym_env")]
        path = os.path.join(REF_ROOT, "robo_rugby", "gym_env", "RR_Constants.py")
        text = open(path).read()
        if preset == "T":
            assert text.count("GAME_MODE = True") == 1
            text = text.replace("GAME_MODE = True", "GAME_MODE = False", 1)
        else:  # D, the duel: one robot per team, one ball of each colour; X: 2 + 1 robots, 2 + 3 balls
            want = dict(NUM_BALL_POS=1, NUM_BALL_NEG=1, NUM_ROBOTS_HAPPY=1, NUM_ROBOTS_GRUMPY=1) if preset == "D" else \
                dict(NUM_BALL_POS=2, NUM_BALL_NEG=3, NUM_ROBOTS_HAPPY=2, NUM_ROBOTS_GRUMPY=1)
            for name, was in (("NUM_BALL_POS", "4 if GAME_MODE else 1"), ("NUM_BALL_NEG", "4 if GAME_MODE else 0"),
                              ("NUM_ROBOTS_HAPPY", "2 if GAME_MODE else 1"), ("NUM_ROBOTS_GRUMPY", "2 if GAME_MODE else 0")):
                line = f"{name} = {was}"
                assert text.count(line) == 1, line
                text = text.replace(line, f"{name} = {want[name]}", 1)
        const = types.ModuleType("robo_rugby.gym_env.RR_Constants")
        const.__file__ = path
        exec(compile(text, path, "exec"), const.__dict__)
        sys.modules["robo_rugby"] = pkg
        sys.modules["robo_rugby.gym_env"] = sub
        sys.modules["robo_rugby.gym_env.RR_Constants"] = const
        sub.RR_Constants = const
    import MyUtils
    import robo_rugby.gym_env.RR_Constants as const
    import robo_rugby.gym_env.RR_EnvBase as base
    import robo_rugby.gym_env.RR_TrashyPhysics as tp
    import robo_rugby.gym_env.RR_Environments as envs
    import robo_rugby.gym_env.RR_Robot as robot
    import robo_rugby.gym_env.RR_Ball as ball
    import robo_rugby.gym_env.RR_Goal as goal
    import robo_rugby.gym_env.RR_ScoreKeepers as sk
    import robo_rugby.gym_env.RR_Observers as obs
    assert bool(const.GAME_MODE) == (preset in ("G", "D", "X"))
    assert const.NUM_ROBOTS_TOTAL == {"G": 4, "T": 1, "D": 2, "X": 3}[preset]
    ns = types.SimpleNamespace(MyUtils=MyUtils, const=const, base=base, tp=tp, envs=envs, robot=robot,
                               ball=ball, goal=goal, sk=sk, obs=obs, preset=preset)
    return ns
